! grtrans_amd/host/pgrtrans.f90
! Drop-in for the reference's module pgrtrans (pgrtrans.f90:1-260): same module name, same
! grtrans_main argument list (85 positional arguments, types as pgrtrans.f90:35-58), same module
! results ivals(nvals+nextra, nx*ny, ncams), ab(2, nx*ny), freqs(nfreq), same del_pgrtrans_data.
! What changes is what happens inside the i=i1..i2 loop: instead of one grtrans_driver call per
! pixel on the host (pgrtrans.f90:196-212) the whole pixel range goes to the MI355X through the
! C ABI (grt_trace_tile), once per (mu0, mdot) pair.
module pgrtrans
  use iso_c_binding
  implicit none

  real(8), allocatable, dimension(:,:,:) :: ivals
  real(8), allocatable, dimension(:,:) :: ab
  real(8), allocatable, dimension(:) :: freqs
  real(8) :: last_ray_steps = 0d0, last_kernel_ms = 0d0    ! extra: what the tracer reported
  ! Error convention.  The reference has none: unsupported input runs into a Fortran STOP (e.g. math.f90:75-79).  So does
  ! this module by default (error stop, exit code 2 or 3).  With soft errors on, grtrans_main instead releases what it
  ! holds, sets last_error (2: refused request, 3: a C-ABI call failed) and last_error_msg, and returns.
  integer, save :: last_error = 0
  character(len=512), save :: last_error_msg = ' '
  logical, save :: soft_errors = .false.
  ! multi-rank runs: a rank that fails must not stop on its own (the others would wait for it in the next RCCL call for ever):
  ! die/die_rc only record the error, every collective of this module carries the error flag (sync_ranks), and all ranks leave
  ! together; the process then stops with the error code unless soft errors are on
  logical, save :: defer_stop = .false.
  ! HARM time-stepping state (module variables of fluid_model_harm.f90:12-16)
  integer, save :: cur_fluid = 0, harm_indf = 0, harm_nt = 1
  real, save :: harm_tstep = 20., harm_toffset = 0.
  real(8), save :: harm_spin = 0d0, harm_scalefac = 1d0
  character(len=256), save :: harm_dfile = ' ', harm_hfile = ' ' 

contains

  subroutine grtrans_main(standard,mumin,mumax,nmu,phi0,spin,&
       uout,uin, rcut, nrotype, gridvals, nn,i1,i2,fname, dt, nt, nload, &
       nmdot, mdotmin, mdotmax,ename, mbh, nfreq, fmin, fmax, muval,&
       gmin, gmax,p1, p2, jetalpha, stype,use_geokerr, nvals, iname,&
       cflag, extra, debug,outfile, fdfile,fhfile,fgfile,fsim,fnt,findf,fnfiles,fjonfix, &
       fnw,fnfreq_tab,fnr,foffset,fdindf,fmagcrit,frspot,fr0spot,fn0spot,ftscl,frscl, &
       fwmin,fwmax,ffmin,ffmax,frmax,fsigt,ffcol,fmdot,fnscl,fnnthscl,fnnthp,fbeta, &
       fbl06,fnp,ftp,frin,frout,fthin,fthout,fphiin,fphiout,fscalefac,sigcut,epcoefindx,epotherargs,nepotherargs)
    use grt_cabi
    use fits_min
    use omp_lib
    integer, intent(in) :: standard,nrotype,nvals,nfreq,nmu,cflag,nt,nmdot,nload,extra,i1,i2,nepotherargs,debug
    logical, intent(in) :: use_geokerr
    real(kind=8), intent(in) :: mumax,mumin,rcut,mbh,uout,uin,fmin,fmax,dt,mdotmin,mdotmax,phi0,muval,gmin,gmax,&
         p1,p2,jetalpha
    character(len=*), intent(in) :: ename,fname,iname,stype
    character(len=*), intent(in) :: outfile
    real(kind=8), dimension(4), intent(in) :: gridvals
    real(kind=8) :: spin
    integer, dimension(3), intent(in) :: nn
    character(len=*), intent(in) :: fdfile,fhfile,fgfile,fsim
    integer, intent(in) :: fnt,findf,fnfiles,fjonfix,fnw,fnfreq_tab,fnr,foffset,fdindf,fmagcrit,fbl06
    real(8), intent(in) :: frspot,fr0spot,fn0spot,ftscl,frscl,fwmin,fwmax,ffmin,ffmax,frmax,fsigt,ffcol,fmdot,&
         fnnthp,fnnthscl,fnscl,fbeta,ftp,fnp,frin,frout,fthin,fthout,fphiin,fphiout,fscalefac,sigcut
    real(8), dimension(nepotherargs), intent(in) :: epotherargs
    integer, dimension(7), intent(in) :: epcoefindx

    integer :: nro, nphi, nup, ncams, nextra, j, l, m, k, cam_index, rc, fluid_id, emis_id, iflag, npix, ntile, dev
    real(8), allocatable :: mdots(:), mu0(:), t0tab(:)
    real(c_float), allocatable :: pixvals(:), pixloc(:), full(:)
    integer(c_int32_t), allocatable :: status(:)
    ! one process per GPU: rank, world size, the pixel ranges of this rank (and, on the root, of every rank)
    integer :: rank, world, local_rank, tilecols, nrng, g, p, nv, force_comm
    integer, allocatable :: rlo(:), rhi(:), plo(:), phi_(:)
    integer(c_int64_t), allocatable :: rlo64(:), rhi64(:)
    integer(c_int64_t), allocatable :: npix_of_rank(:)
    integer(c_int64_t) :: npix_rank, off, pre
    real(8) :: red(2)
    real(8), allocatable :: alpha_all(:), beta_all(:)
    logical :: multi, failed
    type(grt_camera) :: cam
    type(grt_emis) :: em
    type(grt_stats) :: st
    real(8) :: wtime
    character(len=512) :: datadir, path
    integer :: tempi

    last_error = 0; last_error_msg = ' '
    defer_stop = .false.          ! set once this call owns a communicator; never inherited from an earlier call of the process
    nro = nn(1); nphi = nn(2); nup = nn(3); npix = nro*nphi
    if (allocated(ivals) .or. allocated(ab) .or. allocated(freqs)) call del_pgrtrans_data()
    allocate(freqs(nfreq), mu0(nmu), mdots(nmdot))
    ncams = nmdot*nfreq*nmu*nt
    ! log-spaced frequencies / mdots, linear mu0 (pgrtrans.f90:87-104)
    if (nfreq == 1) then
       freqs = (/fmin/)
    else
       freqs = fmin*exp((/(tempi-1,tempi=1,nfreq)/)*log(fmax/fmin)/(nfreq-1))
    end if
    if (nmdot == 1) then
       mdots = (/mdotmin/)
    else
       mdots = mdotmin*exp((/(tempi-1,tempi=1,nmdot)/)*log(mdotmax/mdotmin)/(nmdot-1))
    end if
    if (nmu == 1) then
       mu0 = (/mumin/)
    else
       mu0 = mumin+(mumax-mumin)/(nmu-1)*(/(tempi-1,tempi=1,nmu)/)
    end if
    nextra = 0
    if (extra == 1) then
       if (nup > 1) then
          nextra = 19
       else
          nextra = 7
       end if
    end if

    ! ---- what this build can put on the device; everything else stops loudly (no CPU path) ----
    if (.not. use_geokerr) then
       call die('use_geokerr=F is not supported'); goto 999
    end if
    ! uin never reaches GEOKERR in the reference either (geodesics.f90:71-75); rcut is the outer radius of the polar pixel grid
    ! of nrotype=1 (geokerr_wrapper.f:138-149), gridvals(1) its inner radius
    ! debug=1: geodebug.out of the LAST pixel, mdot, frequency of the run (the reference re-opens the file for every ray, so that is
    ! what a serial run leaves behind); one process only (written after the trace phase, see write_geodebug)
    ! extra=1 with one point per ray: the 7 diagnostics of grtrans_driver.f90:298-311 go through transpol, which reads the
    ! Chandrasekhar table -- loaded only for nvals = 4 (pgrtrans.f90:154), so the reference itself needs BBPOL here
    if (extra == 1 .and. nup <= 1 .and. .not. (trim(ename) == 'BBPOL' .and. nvals == 4)) then
       call die('extra=1 with nup=1 needs ename=BBPOL, nvals=4 (the one-point diagnostics read the Chandrasekhar table)'); goto 999
    end if
    select case (trim(fname))
    case ('FFJET'); fluid_id = GRT_FLUID_FFJET
    case ('SPHACC'); fluid_id = GRT_FLUID_SPHACC
    case ('HARM'); fluid_id = GRT_FLUID_HARM
    case ('HARM3D'); fluid_id = GRT_FLUID_HARM3D
    case ('IHARM'); fluid_id = GRT_FLUID_IHARM
    case ('KORAL3D'); fluid_id = GRT_FLUID_KORAL3D
    case ('KORAL3D_DISK'); fluid_id = GRT_FLUID_KORAL3D_DISK
    case ('THINDISK'); fluid_id = GRT_FLUID_THINDISK
    case default; call die('fluid model '//trim(fname)//' is not on the device yet'); goto 999
    end select
    cur_fluid = fluid_id
    select case (trim(ename))
    case ('POLSYNCHPL'); emis_id = GRT_EMIS_POLSYNCHPL
    case ('SYNCHPL'); emis_id = GRT_EMIS_SYNCHPL
    case ('POLSYNCHTH'); emis_id = GRT_EMIS_POLSYNCHTH
    case ('BB'); emis_id = GRT_EMIS_BB
    case ('BBPOL'); emis_id = GRT_EMIS_BBPOL
    case ('HYBRIDTHPL'); emis_id = GRT_EMIS_HYBRIDTHPL
    case ('HYBRIDTH'); emis_id = GRT_EMIS_HYBRIDTH
    case ('HYBRIDPL'); emis_id = GRT_EMIS_HYBRIDPL
    case ('MAXJUTT'); emis_id = GRT_EMIS_MAXJUTT
    case ('MAXCOMP'); emis_id = GRT_EMIS_MAXCOMP
    case ('POLSYNCHSYMTH'); emis_id = GRT_EMIS_POLSYNCHSYMTH
    case ('SYNCHTHAV'); emis_id = GRT_EMIS_SYNCHTHAV
    case ('SYNCHTHAVNOABS'); emis_id = GRT_EMIS_SYNCHTHAVNOABS
    case ('BREMS'); emis_id = GRT_EMIS_BREMS
    case ('SYNCHTHBREMS'); emis_id = GRT_EMIS_SYNCHTHBREMS
    case ('RHO'); emis_id = GRT_EMIS_RHO
    case default; call die('emissivity '//trim(ename)//' is not on the device yet'); goto 999
    end select
    select case (trim(iname))          ! rad_trans.f90:29-37
    case ('delo'); iflag = 1
    case ('formal'); iflag = 2
    case ('lsoda', 'lsodasph')
       iflag = 0
       ! one stored point per ray: no integrator runs (grtrans_driver.f90:296-300), so the name is immaterial
       if (nup /= 1) then
          call die('iname='//trim(iname)//': LSODA has no device implementation, use formal or delo'); goto 999
       end if
    case default; call die('unknown iname '//trim(iname)); goto 999
    end select

    ! One process per GPU.  The launcher (torchrun, grtrans_batch.run_grtrans(ngpus=N), or any shell loop) starts N copies
    ! with RANK / WORLD_SIZE / LOCAL_RANK set (GRT_RANK / GRT_WORLD_SIZE / GRT_LOCAL_RANK take precedence); every copy reads
    ! the same inputs, traces its share of the pixels i1..i2 and rank 0 assembles ivals / ab and writes the output file.
    rank = env_int('GRT_RANK', env_int('RANK', 0))
    world = env_int('GRT_WORLD_SIZE', env_int('WORLD_SIZE', 1))
    local_rank = env_int('GRT_LOCAL_RANK', env_int('LOCAL_RANK', rank))
    force_comm = env_int('GRT_FORCE_COMM', 0)           ! test hook: the gather path with a one-rank communicator
    multi = (world > 1 .or. force_comm == 1)
    ! share-out of the pixels: contiguous strips of i1..i2 (default) or -- SURVEY 8e, for cameras whose captured or dead
    ! rays cluster -- tiles of GRT_TILE_COLS image columns dealt round-robin to the ranks
    tilecols = env_int('GRT_TILE_COLS', 0)
    if (world < 1 .or. rank < 0 .or. rank >= world) then
       call die('bad RANK / WORLD_SIZE'); goto 999
    end if
    dev = env_int('GRT_DEVICE', merge(local_rank, 0, world > 1))
    rc = grt_init(int(dev, c_int)); if (rc /= 0) then
       call die_rc('grt_init', rc); goto 999
    end if
    if (multi) then
       ! The communicator comes FIRST: from here on a failure of one rank reaches the others through the error flag every
       ! collective carries.  Rendezvous file: GRT_COMM_FILE (the Python launchers make a fresh name per run), else a per-user
       ! name keyed by MASTER_PORT; the run nonce inside it (GRT_COMM_NONCE or the launcher's pid) keeps a file left by a crashed
       ! run from being joined, and rank 0 removes the file once every rank has joined.
       call get_environment_variable('GRT_COMM_FILE', path)
       if (len_trim(path) == 0) then
          call get_environment_variable('MASTER_PORT', path)
          if (len_trim(path) == 0 .and. world > 1) then
             call die('multi-rank run without GRT_COMM_FILE or MASTER_PORT: no rendezvous for the communicator id'); goto 999
          end if
          call get_environment_variable('USER', datadir)
          path = '/tmp/grtrans_amd_comm_'//trim(datadir)//'_'//trim(path)
       end if
       rc = grt_comm_init_file(int(rank, c_int), int(world, c_int), cstr(trim(path)), dble(env_int('GRT_COMM_TIMEOUT', 600)))
       if (rc /= 0) then
          call die_rc('grt_comm_init_file '//trim(path), rc); goto 999
       end if
       defer_stop = .true.
    end if
    call get_environment_variable('GRTRANS_AMD_DATA', datadir)
    if (len_trim(datadir) == 0) datadir = '.'
    path = trim(datadir)//'/polsynchpl_tables.bin'
    rc = grt_load_polsynchpl_tables_file(cstr(path)); if (rc /= 0) then
       call die_rc('grt_load_polsynchpl_tables_file '//trim(path), rc); goto 810
    end if
    if (fluid_id == GRT_FLUID_FFJET) then
       path = trim(fdfile)              ! as load_fluid_model does (fluid.f90:226-228)
       rc = grt_load_ffjet_file(cstr(path))
       if (rc /= 0) then
          path = trim(datadir)//'/'//trim(fdfile)
          rc = grt_load_ffjet_file(cstr(path))
       end if
       if (rc /= 0) then
          call die_rc('grt_load_ffjet_file '//trim(fdfile), rc); goto 810
       end if
    else if (fluid_id == GRT_FLUID_HARM) then
       ! initialize_harm_model(a,ifile,fdfile,fhfile,fnt,findf), fluid.f90:193-194
       ! fnt > 1: that many time slices are loaded and each point reads the one its look-back time selects (slow light)
       rc = grt_load_harm_files_nt(cstr(trim(fdfile)), cstr(trim(fhfile)), int(findf, c_int), int(fnt, c_int), spin)
       if (rc /= 0) then
          call die_rc('grt_load_harm_files '//trim(fdfile)//' '//trim(fhfile), rc); goto 810
       end if
       harm_dfile = fdfile; harm_hfile = fhfile; harm_indf = findf; harm_spin = spin; harm_toffset = 0.
       harm_nt = fnt
       ! tstep = t(1)-t(2) of the loaded dumps when fnt > 1, the module default 20. otherwise (fluid_model_harm.f90:493-500)
       rc = grt_get_harm_tstep(harm_tstep); if (rc /= 0) then
          call die_rc('grt_get_harm_tstep', rc); goto 810
       end if
    else if (fluid_id == GRT_FLUID_HARM3D) then
       harm_tstep = 20.
       ! initialize_harm3d_model(a,ifile,fdfile,fhfile,fgfile,fnt,findf), fluid.f90:195-196; one loaded slice
       if (fnt /= 1 .or. nload /= 1) then
          ! read_harm3d_data_file never assigns the dump time on its binary branch (fluid_model_harm3d.f90:383, 402-449): the
          ! reference's tstep is whatever the stack held ("WARNING -- Time step changes over dumps!")
          call die('HARM3D with several loaded slices (fnt, nload > 1): undefined in the reference (dump time never read), refused'); goto 810
       end if
       rc = grt_load_harm3d_files(cstr(trim(fdfile)), cstr(trim(fhfile)), int(findf, c_int), spin)
       if (rc /= 0) then
          call die_rc('grt_load_harm3d_files '//trim(fdfile)//' '//trim(fhfile), rc); goto 810
       end if
       harm_dfile = fdfile; harm_hfile = fhfile; harm_indf = findf; harm_spin = spin; harm_toffset = 0.
       harm_nt = 1
    else if (fluid_id == GRT_FLUID_KORAL3D .or. fluid_id == GRT_FLUID_KORAL3D_DISK) then
       ! initialize_koral3d_model(a,ifile,.false.,fdfile,fhfile,fnt,findf,fscalefac,...), fluid.f90:205-210.  fnt > 1: that many
       ! slices are loaded and each point reads the one its look-back time selects (fluid_model_koral3d.f90:126-128, 212-228)
       rc = grt_load_koral3d_files_nt(cstr(trim(fdfile)), cstr(trim(fhfile)), int(findf, c_int), int(max(fnt, 1), c_int), spin, fscalefac)
       if (rc /= 0) then
          call die_rc('grt_load_koral3d_files '//trim(fdfile)//' '//trim(fhfile), rc); goto 810
       end if
       harm_dfile = fdfile; harm_hfile = fhfile; harm_indf = findf; harm_spin = spin; harm_toffset = 0.
       harm_nt = max(fnt, 1); harm_scalefac = fscalefac
       ! tstep = t(1)-t(2) of the loaded dumps when fnt > 1, the module default 20. otherwise (fluid_model_koral3d.f90:23, 617-619)
       rc = grt_get_koral3d_tstep(harm_tstep); if (rc /= 0) then
          call die_rc('grt_get_koral3d_tstep', rc); goto 810
       end if
    else if (fluid_id == GRT_FLUID_IHARM) then
       harm_tstep = 20.
       ! initialize_iharm_model(a,ifile,fdfile,fhfile,fgfile,fnt,findf), fluid.f90:199-200; one loaded slice, metric 0
       if (fnt /= 1 .or. nload /= 1) then
          ! load_iharm_data stores the HEADER's time for every slice (fluid_model_iharm.f90:745): tstep = t(1)-t(2) = 0 (:778)
          call die('IHARM with several loaded slices (fnt, nload > 1): undefined in the reference (tstep = 0), refused'); goto 810
       end if
       rc = grt_load_iharm_files(cstr(trim(fdfile)), cstr(trim(fhfile)), int(findf, c_int), spin)
       if (rc /= 0) then
          call die_rc('grt_load_iharm_files '//trim(fdfile)//' '//trim(fhfile), rc); goto 810
       end if
       harm_dfile = fdfile; harm_hfile = fhfile; harm_indf = findf; harm_spin = spin; harm_toffset = 0.
       harm_nt = 1
    else if (fluid_id == GRT_FLUID_THINDISK) then
       ! init_thindisk(real(a),ifile,real(fargs%mdot),real(fargs%mbh),real(fargs%rin),real(fargs%rout)), fluid.f90:226-227
       rc = grt_set_thindisk(fmdot, frin, frout); if (rc /= 0) then
          call die_rc('grt_set_thindisk', rc); goto 810
       end if
       if (nup == 1 .and. nvals == 4) then
          ! load_chandra_tab24 (pgrtrans.f90:154): ch24_vals.txt from the working directory, as the reference; else the shipped copy
          rc = grt_load_chandra_file(cstr('ch24_vals.txt'))
          if (rc /= 0) then
             path = trim(datadir)//'/ch24_vals.txt'
             rc = grt_load_chandra_file(cstr(path))
          end if
          if (rc /= 0) then
             call die_rc('grt_load_chandra_file ch24_vals.txt', rc); goto 810
          end if
       end if
    else if (fluid_id == GRT_FLUID_SPHACC) then
       ! the reference holds these tables as DATA statements (fluid_model_sphacc.f90:40-448)
       path = trim(datadir)//'/sphacc_tables.bin'
       rc = grt_load_sphacc_tables_file(cstr(path)); if (rc /= 0) then
          call die_rc('grt_load_sphacc_tables_file '//trim(path), rc); goto 810
       end if
    end if

    if (rank == 0) write(6,*) 'welcome to grtrans! (MI355X per-pixel path)'
    nv = nvals + nextra
    allocate(ivals(nv, npix, ncams)); ivals = 0d0
    allocate(ab(2, npix)); ab = 0d0
    ! this rank's pixel ranges (one range = one grt_trace_tile call); with one process it is i1..i2 itself
    call pixel_ranges(rank, world, i1, i2, nphi, tilecols, nrng, rlo, rhi)
    npix_rank = 0
    allocate(rlo64(max(nrng, 1)), rhi64(max(nrng, 1)))
    do g = 1, nrng
       npix_rank = npix_rank + (rhi(g) - rlo(g) + 1)
       rlo64(g) = rlo(g); rhi64(g) = rhi(g)
    end do
    ntile = int(max(npix_rank, 1_c_int64_t))
    if (multi) then
       rc = grt_stage_reset(int(ncams, c_int), int(nv, c_int), npix_rank); if (rc /= 0) then
          call die_rc('grt_stage_reset', rc)
       end if
    end if
    ! ---- sync point: every rank arrives here whether its loading phase failed or not; all leave together on any failure ----
810 continue
    if (multi) then
       red(1) = 0d0
       call sync_ranks(red(1), failed)
       if (failed) goto 999
    else if (last_error /= 0) then
       goto 999
    end if
    allocate(pixvals(int(nv,8)*ntile*nfreq), pixloc(2*ntile), status(ntile))
    last_ray_steps = 0d0; last_kernel_ms = 0d0
    do j = 1, nmu
       rc = grt_make_camera(cam, int(standard, c_int), gridvals, int(nn, c_int), spin, mu0(j), uout, phi0)
       if (rc /= 0) then
          call die_rc('grt_make_camera', rc); goto 820
       end if
       if (nrotype == 1) then
          rc = grt_set_camera_polar(cam, rcut)
          if (rc /= 0) then
             call die_rc('grt_set_camera_polar', rc); goto 820
          end if
       end if
       if (nload > 1) then
          ! pgrtrans.f90:172-189: time origin of every ray from a one-point geodesic, normalised over the image
          ! (over all ranks: two max-reductions of statistics, not a data-path collective);
          ! initialize_geo_tabs (geodesics.f90:120) leaves the mu-fill segment switched off for what follows
          cam%mufill = 0; cam%kext = 0; cam%next = 0
          allocate(t0tab(max(npix_rank, 1_c_int64_t)))
          off = 0
          do g = 1, nrng
             rc = grt_compute_t0(cam, int(rlo(g), c_int64_t), int(rhi(g), c_int64_t), t0tab(off+1:))
             if (rc /= 0) then
                call die_rc('grt_compute_t0', rc); goto 820
             end if
             off = off + (rhi(g) - rlo(g) + 1)
          end do
          red(1) = -huge(1d0)
          if (npix_rank > 0) red(1) = maxval(t0tab(1:npix_rank))
          if (multi) then
             call sync_ranks(red(1), failed)
             if (failed) goto 999
          end if
          where (t0tab > 0.)
             t0tab = t0tab - red(1)
          elsewhere
             t0tab = 0.
          end where
          red(1) = -huge(1d0)
          if (npix_rank > 0) red(1) = -minval(t0tab(1:npix_rank))
          if (multi) then
             call sync_ranks(red(1), failed)
             if (failed) goto 999
          end if
          t0tab = t0tab + red(1)
       end if
       wtime = omp_get_wtime()
       do l = 1, nt
          if (l > 1) then
             call advance_fluid_timestep(dt)             ! pgrtrans.f90:211 (after frame l-1)
             if (last_error /= 0) goto 820
          end if
          ! every range of this rank in ONE call (grt_trace_ranges): one set of launches fills the GPU where a call per column
          ! tile would not; the rows of pixvals follow the ranges in order, like the rank-level image the gather sends
          if (nrng > 0) then
             if (nload > 1) then
                rc = grt_set_t0(t0tab, npix_rank); if (rc /= 0) then
                   call die_rc('grt_set_t0', rc); goto 820
                end if
             end if
             ! one geodesic serves every m = 1..nmdot (grtrans_driver.f90:134): after the first mdot the library re-uses the
             ! resident geodesic record of this camera and these ranges and launches only the transfer kernels
             do m = 1, nmdot
                em%mbh = mbh; em%nfac = ftscl; em%bfac = frscl          ! pgrtrans.f90:125-128
                em%gmin = dble(real(gmin)); em%gmax = dble(real(gmax)); em%p1 = dble(real(p1))  ! emis.f90:32-39
                em%muval = muval; em%etype = emis_id; em%coefindx = epcoefindx
                if (nepotherargs < 2 .or. nepotherargs > 8) then
                   call die('epotherargs = (delta, 1..7 weights)'); goto 820
                end if
                em%nother = nepotherargs; em%otherargs = 0d0; em%otherargs(1:nepotherargs) = epotherargs   ! pgrtrans.f90:119-120
                em%jetalpha = jetalpha; em%sigcut = sigcut               ! sparams%jetalphaval, %sigcut (pgrtrans.f90:129-136)
                em%mdot = mdots(m); em%gminval = gmin                   ! sparams%mdot, %gminval (pgrtrans.f90:127-131)
                rc = grt_trace_ranges(cam, int(nrng, c_int), rlo64, rhi64, fluid_id, em, int(nfreq, c_int), freqs, &
                     int(iflag, c_int), int(nvals, c_int), int(nextra, c_int), pixvals, pixloc, status, st)
                if (rc /= 0) then
                   call die_rc('grt_trace_ranges', rc); goto 820
                end if
                last_ray_steps = last_ray_steps + dble(st%ray_steps); last_kernel_ms = last_kernel_ms + st%ms_total
                ! camera index (l-1)*nfreq*nparams+(m-1)*nfreq+k with l -> (j-1)*nt+l (grtrans_driver.f90:324)
                cam_index = ((j-1)*nt + l - 1)*nfreq*nmdot + (m-1)*nfreq
                if (multi) then
                   ! the rank's pixels stay in HBM: device-to-device into this rank's image, gathered once at the end
                   rc = grt_stage_resident(int(cam_index, c_int), int(nfreq, c_int), npix_rank, 0_c_int64_t)
                   if (rc /= 0) then
                      call die_rc('grt_stage_resident', rc); goto 820
                   end if
                else
                   off = 0
                   do g = 1, nrng
                      ntile = rhi(g) - rlo(g) + 1
                      do k = 1, nfreq
                         ivals(:, rlo(g):rhi(g), cam_index + k) = reshape(dble(pixvals(((k-1)*npix_rank + off)*nv + 1 : &
                              ((k-1)*npix_rank + off + ntile)*nv)), (/nv, ntile/))
                      end do
                      off = off + ntile
                   end do
                end if
             end do
          end if
       end do
       if (allocated(t0tab)) deallocate(t0tab)
       if (rank == 0) write(6,*) 'grtrans wall time elapsed: ', omp_get_wtime() - wtime
    end do
    if (debug /= 0) then
       if (world > 1) then
          call die('debug=1 (geodebug.out) with more than one rank: run the pixel of interest with one process'); goto 820
       end if
       em%mdot = mdots(nmdot)
       call write_geodebug(cam, int(i2, c_int64_t), fluid_id, em, freqs(nfreq), iflag, nvals)
       if (last_error /= 0) goto 820
    end if
    ! ---- sync point: end of the trace phase; no rank enters the gather unless every rank has its image ----
820 continue
    if (multi) then
       red(1) = 0d0
       call sync_ranks(red(1), failed)
       if (failed) goto 999
    else if (last_error /= 0) then
       goto 999
    end if
    ! ---- assembly of ivals on rank 0 (pgrtrans.f90:221-242): one RCCL gather of the ranks' fp32 images over xGMI ----
    if (multi) then
       allocate(npix_of_rank(world))
       do p = 0, world - 1
          call pixel_ranges(p, world, i1, i2, nphi, tilecols, k, plo, phi_)
          npix_of_rank(p+1) = 0
          do g = 1, k
             npix_of_rank(p+1) = npix_of_rank(p+1) + (phi_(g) - plo(g) + 1)
          end do
       end do
       if (rank == 0) then
          allocate(full(max(int(nv,8)*sum(npix_of_rank)*ncams, 1_8)))
       else
          allocate(full(1))
       end if
       rc = grt_gather_stage(npix_of_rank, 0_c_int, full); if (rc /= 0) then
          call die_rc('grt_gather_stage', rc); goto 999
       end if
       red(1) = last_ray_steps; red(2) = 0d0
       rc = grt_comm_allreduce(red, 1_c_int, 0_c_int); if (rc /= 0) then
          call die_rc('grt_comm_allreduce', rc); goto 999
       end if
       last_ray_steps = red(1)
       red(1) = last_kernel_ms
       rc = grt_comm_allreduce(red, 1_c_int, 1_c_int); if (rc /= 0) then
          call die_rc('grt_comm_allreduce', rc); goto 999
       end if
       last_kernel_ms = red(1)
       if (rank == 0) then
          ! full = [ncams][all ranks' pixels, rank after rank, each rank's ranges in order][nv]
          pre = 0
          do p = 0, world - 1
             call pixel_ranges(p, world, i1, i2, nphi, tilecols, k, plo, phi_)
             off = 0
             do g = 1, k
                ntile = phi_(g) - plo(g) + 1
                do m = 1, ncams
                   ivals(:, plo(g):phi_(g), m) = reshape(dble(full((int(m-1,8)*sum(npix_of_rank) + pre + off)*nv + 1 : &
                        (int(m-1,8)*sum(npix_of_rank) + pre + off + ntile)*nv)), (/nv, ntile/))
                end do
                off = off + ntile
             end do
             pre = pre + npix_of_rank(p+1)
          end do
       end if
       deallocate(full, npix_of_rank)
       rc = grt_comm_free()
    end if
    if (rank == 0) write(6,*) 'ray-steps: ', last_ray_steps, ' device ms: ', last_kernel_ms, ' ranks: ', world
    ! c(1)%pixloc (pgrtrans.f90:227): alpha, beta of the traced pixels i1..i2 as REAL(4) (camera.f90:36); pure arithmetic
    ntile = i2 - i1 + 1
    allocate(alpha_all(ntile), beta_all(ntile))
    rc = grt_make_camera(cam, int(standard, c_int), gridvals, int(nn, c_int), spin, mu0(nmu), uout, phi0)
    if (nrotype == 1) rc = grt_set_camera_polar(cam, rcut)
    rc = grt_camera_pixels_ab(cam, int(i1, c_int64_t), int(i2, c_int64_t), alpha_all, beta_all)
    if (rc /= 0) then
       call die_rc('grt_camera_pixels', rc); goto 999
    end if
    ab(1, i1:i2) = dble(real(alpha_all)); ab(2, i1:i2) = dble(real(beta_all))
    deallocate(alpha_all, beta_all)
    if (rank /= 0) then          ! only the root holds the assembled image and writes the file
       deallocate(pixvals, pixloc, status, mu0, mdots)
       rc = grt_free()
       defer_stop = .false.      ! (a later single-process call in the same host process must stop where it fails again)
       return
    end if
    if (len_trim(outfile) > 0 .and. cflag /= 1) then
       do m = 1, ncams
          call write_camera(trim(outfile), m, nro, nphi, nvals+nextra, freqs(1+mod(m-1,nfreq)), ab, ivals(:,:,m), cflag)
       end do
    else if (len_trim(outfile) > 0) then
       ! cflag=1: the FITS layout of kwrite_raytrace_camera (camera.f90:185-319) through the built-in emitter
       do m = 1, ncams
          if (m == 1) then
             call fits_open_new(12, trim(outfile))
             call fits_primary(12, real(reshape(ab, (/2*npix/))))
          else
             call fits_open_append(12, trim(outfile))
          end if
          call fits_image_begin(npix*(nvals+nextra))
          call fits_key_d('nx', dble(real(nro)), '# x pixels')
          call fits_key_d('ny', dble(real(nphi)), '# y pixels')
          call fits_key_d('nu', dble(real(freqs(1+mod(m-1,nfreq)))), 'Frequency (Hz)')
          call fits_key_i('standard', standard, 'standard'); call fits_key_d('mumin', mumin, 'mumin')
          call fits_key_d('mumax', mumax, 'mumax'); call fits_key_i('nmu', nmu, 'nmu'); call fits_key_d('phi0', phi0, 'phi0')
          call fits_key_d('spin', spin, 'spin'); call fits_key_d('uout', uout, 'uout'); call fits_key_d('uin', uin, 'uin')
          call fits_key_d('rcut', rcut, 'rcut'); call fits_key_i('nrotype', nrotype, 'nrotype')
          call fits_key_d('gridvals(1)', gridvals(1), 'gridvals(1)'); call fits_key_d('gridvals(2)', gridvals(2), 'gridvals(2)')
          call fits_key_d('gridvals(3)', gridvals(3), 'gridvals(3)'); call fits_key_d('gridvals(4)', gridvals(4), 'gridvals(4)')
          call fits_key_i('nn(1)', nn(1), 'nn(1)'); call fits_key_i('nn(2)', nn(2), 'nn(2)'); call fits_key_i('nn(3)', nn(3), 'nn(3)')
          call fits_key_s('fname', trim(fname), 'fname'); call fits_key_d('dt', dt, 'dt'); call fits_key_i('nt', nt, 'nt')
          call fits_key_i('nload', nload, 'nload'); call fits_key_i('nmdot', nmdot, 'nmdot')
          call fits_key_d('mdotmin', mdotmin, 'mdotmin'); call fits_key_d('mdotmax', mdotmax, 'mdotmax')
          call fits_key_s('ename', trim(ename), 'ename'); call fits_key_d('mbh', mbh, 'mbh'); call fits_key_i('nfreq', nfreq, 'nfreq')
          call fits_key_d('fmin', fmin, 'fmin'); call fits_key_d('fmax', fmax, 'fmax'); call fits_key_d('muval', muval, 'muval')
          call fits_key_d('gmin', gmin, 'gmin'); call fits_key_d('gmax', gmax, 'gmax'); call fits_key_d('p1', p1, 'p1')
          call fits_key_d('p2', p2, 'p2'); call fits_key_d('jetalpha', jetalpha, 'jetalpha'); call fits_key_s('stype', trim(stype), 'stype')
          call fits_key_s('use_geokerr', merge('T', 'F', use_geokerr), 'use_geokerr'); call fits_key_i('nvals', nvals, 'nvals')
          call fits_key_s('iname', trim(iname), 'iname'); call fits_key_i('cflag', cflag, 'cflag'); call fits_key_i('extra', extra, 'extra')
          call fits_image_end(12, real(reshape(ivals(:,:,m), (/npix*(nvals+nextra)/))))
          call fits_close(12)
       end do
    end if
    deallocate(pixvals, pixloc, status, mu0, mdots)
    rc = grt_free()
    defer_stop = .false.
    return
    ! soft-error exit (pgrtrans_set_soft_errors: callers that must survive a refusal, e.g. the Python harness): everything
    ! this call allocated is released, last_error / last_error_msg say why
999 continue
    if (allocated(pixvals)) deallocate(pixvals)
    if (allocated(pixloc)) deallocate(pixloc)
    if (allocated(status)) deallocate(status)
    if (allocated(mu0)) deallocate(mu0)
    if (allocated(mdots)) deallocate(mdots)
    if (allocated(t0tab)) deallocate(t0tab)
    if (allocated(full)) deallocate(full)
    if (allocated(npix_of_rank)) deallocate(npix_of_rank)
    call del_pgrtrans_data()
    rc = grt_comm_free()
    rc = grt_free()
    if (defer_stop) then
       defer_stop = .false.
       if (.not. soft_errors) then
          if (last_error == 2) error stop 2
          error stop 3
       end if
    end if
  end subroutine grtrans_main

  ! max over the ranks of x, and of the "this rank has failed" flag: every collective of the host carries the flag, so that a
  ! rank that failed (its die/die_rc only recorded the error) and the ranks that did not leave grtrans_main together.
  subroutine sync_ranks(x, failed)
    use grt_cabi
    real(8), intent(inout) :: x
    logical, intent(out) :: failed
    real(8) :: r2(2)
    integer :: rc
    r2(1) = x; r2(2) = merge(1d0, 0d0, last_error /= 0)
    rc = grt_comm_allreduce(r2, 2_c_int, 1_c_int)
    if (rc /= 0) then
       call die_rc('grt_comm_allreduce', rc)
       failed = .true.
       return
    end if
    x = r2(1)
    failed = r2(2) > 0d0
    if (failed .and. last_error == 0) then
       last_error = 4; last_error_msg = 'another rank failed'
       write(0,*) 'grtrans_amd: another rank failed, leaving with it'
    end if
  end subroutine sync_ranks

  ! Binary camera file, one block per camera: the record structure of camera.f90:321-340
  ! (nx,ny,nvals / nkey / nkey x real(4) records / pixloc / pixvals), Fortran sequential unformatted.
  subroutine write_camera(outfile, cnum, nx, ny, nv, freq, abv, iv, cflag)
    character(len=*), intent(in) :: outfile
    integer, intent(in) :: cnum, nx, ny, nv, cflag
    real(8), intent(in) :: freq, abv(:,:), iv(:,:)
    integer :: k
    real(4) :: kvals(3)
    if (cnum == 1) then
       open(unit=12, file=outfile, form='unformatted', status='replace')
    else
       open(unit=12, file=outfile, form='unformatted', position='append', status='old')
    end if
    kvals = (/real(nx), real(ny), real(freq)/)
    write(12) nx, ny, nv
    write(12) 3
    do k = 1, 3
       write(12) kvals(k)
    end do
    write(12) real(abv)
    write(12) real(iv)
    close(12)
  end subroutine write_camera

  ! geodebug.out in the reference's block order (grtrans_driver.f90:94-110, 341-421; read_geodebug_file.py reads one block per
  ! line, which is how it is written here -- the reference's list-directed WRITE wraps long records, compiler permitting)
  subroutine write_geodebug(cam, pixel, fluid_id, em, freq, iflag, nvals)
    use grt_cabi
    type(grt_camera), intent(in) :: cam
    integer(c_int64_t), intent(in) :: pixel
    integer, intent(in) :: fluid_id, iflag, nvals
    type(grt_emis), intent(in) :: em
    real(8), intent(in) :: freq
    real(c_double), allocatable :: rows(:)
    real(c_double) :: header(11), fac
    integer(c_int32_t) :: n
    integer :: maxpts, rc, b, np
    maxpts = cam%nup + cam%next
    allocate(rows(64*maxpts))
    rc = grt_debug_ray(cam, pixel, int(fluid_id, c_int), em, freq, int(iflag, c_int), int(nvals, c_int), int(maxpts, c_int), n, header, rows, fac)
    if (rc /= 0) then
       call die_rc('grt_debug_ray (debug=1 serves FFJET+POLSYNCHPL, SPHACC+POLSYNCHTH, HARM+POLSYNCHTH with nvals=4)', rc)
       return
    end if
    np = abs(n)
    open(unit=9, file='geodebug.out', form='formatted', status='replace')
    write(9, '(i0,1x,i0,9(1x,es24.16e3))') np, 18, header(3:11)
    if (np >= 2) then
       do b = 1, 49
          write(9, '(*(es24.16e3,1x))') rows((b-1)*maxpts+1 : (b-1)*maxpts+np)
       end do
       ! aahat(npts,3): one block, column after column
       write(9, '(*(es24.16e3,1x))') rows(49*maxpts+1 : 49*maxpts+np), rows(50*maxpts+1 : 50*maxpts+np), rows(51*maxpts+1 : 51*maxpts+np)
       do b = 53, 64
          write(9, '(*(es24.16e3,1x))') rows((b-1)*maxpts+1 : (b-1)*maxpts+np)
       end do
       write(9, '(es24.16e3)') fac
    end if
    close(9)
    deallocate(rows)
  end subroutine write_geodebug

  subroutine del_pgrtrans_data()
    if (allocated(ivals)) deallocate(ivals)
    if (allocated(ab)) deallocate(ab)
    if (allocated(freqs)) deallocate(freqs)
  end subroutine del_pgrtrans_data

  ! advance_fluid_timestep (fluid.f90:246-262): only HARM-family models step (advance_harm_timestep,
  ! fluid_model_harm.f90:511-521: REAL(4) tstep -- the dumps' own spacing when several slices are loaded, 20. otherwise --
  ! and toffset kept across calls; advance_koral3d_timestep, fluid_model_koral3d.f90:635-645, is the same statement for
  ! statement); the other device models are static
  subroutine advance_fluid_timestep(dtobs)
    use grt_cabi
    real(8), intent(in) :: dtobs
    integer :: nupdate, rc
    if (cur_fluid /= GRT_FLUID_HARM .and. cur_fluid /= GRT_FLUID_HARM3D .and. cur_fluid /= GRT_FLUID_IHARM .and. &
        cur_fluid /= GRT_FLUID_KORAL3D .and. cur_fluid /= GRT_FLUID_KORAL3D_DISK) return
    nupdate = floor(dtobs / harm_tstep)
    harm_toffset = harm_toffset + dtobs - harm_tstep*nupdate
    nupdate = nupdate + floor(harm_toffset/harm_tstep)
    harm_toffset = harm_toffset - floor(harm_toffset/harm_tstep)*harm_tstep
    harm_indf = harm_indf + nupdate
    write(6,*) 'advance harm timestep: ', dtobs, harm_tstep, harm_toffset, harm_indf, nupdate
    ! nupdate = 0 leaves the slice as it is (update_harm_data(0) loads nothing)
    if (nupdate > 0 .and. cur_fluid == GRT_FLUID_IHARM) then
       rc = grt_load_iharm_files(cstr(trim(harm_dfile)), cstr(trim(harm_hfile)), int(harm_indf, c_int), harm_spin)
       if (rc /= 0) call die_rc('grt_load_iharm_files (advance) '//trim(harm_dfile), rc)
    else if (nupdate > 0 .and. (cur_fluid == GRT_FLUID_KORAL3D .or. cur_fluid == GRT_FLUID_KORAL3D_DISK)) then
       rc = grt_load_koral3d_files_nt(cstr(trim(harm_dfile)), cstr(trim(harm_hfile)), int(harm_indf, c_int), int(harm_nt, c_int), &
                                      harm_spin, harm_scalefac)
       if (rc /= 0) call die_rc('grt_load_koral3d_files (advance) '//trim(harm_dfile), rc)
    else if (nupdate > 0 .and. cur_fluid == GRT_FLUID_HARM3D) then
       rc = grt_load_harm3d_files(cstr(trim(harm_dfile)), cstr(trim(harm_hfile)), int(harm_indf, c_int), harm_spin)
       if (rc /= 0) call die_rc('grt_load_harm3d_files (advance) '//trim(harm_dfile), rc)
    else if (nupdate > 0) then
       rc = grt_load_harm_files_nt(cstr(trim(harm_dfile)), cstr(trim(harm_hfile)), int(harm_indf, c_int), int(harm_nt, c_int), harm_spin)
       if (rc /= 0) call die_rc('grt_load_harm_files (advance) '//trim(harm_dfile), rc)
    end if
  end subroutine advance_fluid_timestep

  integer function env_int(name, default)
    character(len=*), intent(in) :: name
    integer, intent(in) :: default
    character(len=64) :: v
    integer :: ios
    env_int = default
    call get_environment_variable(name, v)
    if (len_trim(v) > 0) then
       read(v, *, iostat=ios) env_int
       if (ios /= 0) env_int = default
    end if
  end function env_int

  ! Pixel ranges (1-based inclusive, the reference's own i1..i2 key, read_inputs.f90:8-9) of rank r out of `world`.
  ! tilecols <= 0: one contiguous near-equal strip of i1..i2 per rank.  tilecols > 0: the image columns (alpha index; a
  ! column = nphi consecutive pixels, geokerr_wrapper.f:258-259) in tiles of tilecols, dealt round-robin (one range per
  ! tile, also with a single rank).  Every pixel of i1..i2 belongs to exactly one range of exactly one rank.
  subroutine pixel_ranges(r, world, i1, i2, nphi, tilecols, n, lo, hi)
    integer, intent(in) :: r, world, i1, i2, nphi, tilecols
    integer, intent(out) :: n
    integer, allocatable, intent(out) :: lo(:), hi(:)
    integer :: ntot, base, rem, t, t1, t2, a, b, cnt
    ntot = i2 - i1 + 1
    if (tilecols <= 0) then
       base = ntot / world; rem = mod(ntot, world)
       a = i1 + r*base + min(r, rem)
       b = a + base - 1
       if (r < rem) b = b + 1
       if (b >= a) then
          n = 1; allocate(lo(1), hi(1)); lo(1) = a; hi(1) = b
       else
          n = 0; allocate(lo(0), hi(0))
       end if
       return
    end if
    t1 = (i1 - 1) / (tilecols*nphi); t2 = (i2 - 1) / (tilecols*nphi)
    cnt = 0
    do t = t1, t2
       if (mod(t, world) == r) cnt = cnt + 1
    end do
    allocate(lo(cnt), hi(cnt))
    n = 0
    do t = t1, t2
       if (mod(t, world) /= r) cycle
       a = max(i1, t*tilecols*nphi + 1); b = min(i2, (t+1)*tilecols*nphi)
       if (b < a) cycle
       n = n + 1; lo(n) = a; hi(n) = b
    end do
  end subroutine pixel_ranges

  ! test access to the share-out (tests/test_host_and_abi.py compares it with grtrans_amd/tiling.py)
  subroutine pgrtrans_pixel_ranges(r, world, i1, i2, nphi, tilecols, nmax, n, lo, hi) bind(C, name='pgrtrans_pixel_ranges')
    integer(c_int), value :: r, world, i1, i2, nphi, tilecols, nmax
    integer(c_int), intent(out) :: n, lo(*), hi(*)
    integer, allocatable :: a(:), b(:)
    integer :: nn_
    call pixel_ranges(r, world, i1, i2, nphi, tilecols, nn_, a, b)
    n = nn_
    lo(1:min(nn_, nmax)) = a(1:min(nn_, nmax)); hi(1:min(nn_, nmax)) = b(1:min(nn_, nmax))
  end subroutine pgrtrans_pixel_ranges

  integer function grt_camera_pixels_ab(cam, i1, i2, alpha, beta) result(rc)
    use grt_cabi
    type(grt_camera), intent(in) :: cam
    integer(c_int64_t), intent(in) :: i1, i2
    real(8), intent(out) :: alpha(*), beta(*)
    rc = grt_camera_pixels(cam, i1, i2, alpha, beta, c_null_ptr, c_null_ptr, c_null_ptr, c_null_ptr, c_null_ptr, c_null_ptr, &
         c_null_ptr, c_null_ptr)
  end function grt_camera_pixels_ab

  subroutine die(msg)
    character(len=*), intent(in) :: msg
    write(0,*) 'grtrans_amd: ', trim(msg)
    if (soft_errors .or. defer_stop) then
       last_error = 2; last_error_msg = msg
       return
    end if
    error stop 2
  end subroutine die

  subroutine die_rc(what, rc)
    character(len=*), intent(in) :: what
    integer, intent(in) :: rc
    write(0,*) 'grtrans_amd: ', trim(what), ' failed, code ', rc
    if (soft_errors .or. defer_stop) then
       last_error = 3
       write(last_error_msg, '(a,a,i0)') trim(what), ' failed, code ', rc
       return
    end if
    error stop 3
  end subroutine die_rc

  ! ---- what an f2py wrapper of this module gives a Python caller: the module arrays, by copy (ctypes cannot read a
  ! Fortran array descriptor) ----
  subroutine pgrtrans_shape(nv, npix, ncams, nfreq) bind(C, name='pgrtrans_shape')
    integer(c_int), intent(out) :: nv, npix, ncams, nfreq
    nv = 0; npix = 0; ncams = 0; nfreq = 0
    if (allocated(ivals)) then
       nv = size(ivals, 1); npix = size(ivals, 2); ncams = size(ivals, 3)
    end if
    if (allocated(freqs)) nfreq = size(freqs)
  end subroutine pgrtrans_shape

  subroutine pgrtrans_copy(ivals_out, ab_out, freqs_out) bind(C, name='pgrtrans_copy')
    real(c_double), intent(out) :: ivals_out(*), ab_out(*), freqs_out(*)
    if (allocated(ivals)) ivals_out(1:size(ivals)) = reshape(ivals, (/size(ivals)/))
    if (allocated(ab)) ab_out(1:size(ab)) = reshape(ab, (/size(ab)/))
    if (allocated(freqs)) freqs_out(1:size(freqs)) = freqs
  end subroutine pgrtrans_copy

  subroutine pgrtrans_set_soft_errors(flag) bind(C, name='pgrtrans_set_soft_errors')
    integer(c_int), value :: flag
    soft_errors = (flag /= 0)
    last_error = 0; last_error_msg = ' '
  end subroutine pgrtrans_set_soft_errors

  subroutine pgrtrans_last_error(code, msg, nmsg) bind(C, name='pgrtrans_last_error')
    integer(c_int), intent(out) :: code
    integer(c_int), value :: nmsg
    character(kind=c_char), intent(out) :: msg(*)
    integer :: i
    code = last_error
    do i = 1, min(nmsg - 1, len_trim(last_error_msg))
       msg(i) = last_error_msg(i:i)
    end do
    msg(min(nmsg, len_trim(last_error_msg) + 1)) = c_null_char
  end subroutine pgrtrans_last_error

  subroutine pgrtrans_stats(ray_steps, kernel_ms) bind(C, name='pgrtrans_stats')
    real(c_double), intent(out) :: ray_steps, kernel_ms
    ray_steps = last_ray_steps; kernel_ms = last_kernel_ms
  end subroutine pgrtrans_stats

end module pgrtrans
