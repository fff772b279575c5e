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
  ! HARM time-stepping state (module variables of fluid_model_harm.f90:12-16)
  integer, save :: cur_fluid = 0, harm_indf = 0, harm_nt = 1
  real, save :: harm_tstep = 20., harm_toffset = 0.
  real(8), save :: harm_spin = 0d0
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
    real(c_float), allocatable :: pixvals(:), pixloc(:)
    integer(c_int32_t), allocatable :: status(:)
    type(grt_camera) :: cam
    type(grt_emis) :: em
    type(grt_stats) :: st
    real(8) :: wtime
    character(len=512) :: datadir, path
    integer :: tempi

    nro = nn(1); nphi = nn(2); nup = nn(3); npix = nro*nphi
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
    if (.not. use_geokerr) call die('use_geokerr=F is not supported')
    if (debug /= 0) call die('debug=1 (geodebug.out) is not supported; use grt_set_record_steps / grt_download_steps')
    if (extra == 1 .and. nup <= 1) call die('extra=1 with nup=1 is not supported yet')
    select case (trim(fname))
    case ('FFJET'); fluid_id = GRT_FLUID_FFJET
    case ('SPHACC'); fluid_id = GRT_FLUID_SPHACC
    case ('HARM'); fluid_id = GRT_FLUID_HARM
    case ('HARM3D'); fluid_id = GRT_FLUID_HARM3D
    case ('IHARM'); fluid_id = GRT_FLUID_IHARM
    case ('KORAL3D'); fluid_id = GRT_FLUID_KORAL3D
    case ('KORAL3D_DISK'); fluid_id = GRT_FLUID_KORAL3D_DISK
    case ('THINDISK'); fluid_id = GRT_FLUID_THINDISK
    case default; call die('fluid model '//trim(fname)//' is not on the device yet')
    end select
    cur_fluid = fluid_id
    select case (trim(ename))
    case ('POLSYNCHPL'); emis_id = GRT_EMIS_POLSYNCHPL
    case ('SYNCHPL'); emis_id = GRT_EMIS_SYNCHPL
    case ('POLSYNCHTH'); emis_id = GRT_EMIS_POLSYNCHTH
    case ('BB'); emis_id = GRT_EMIS_BB
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
    case default; call die('emissivity '//trim(ename)//' is not on the device yet')
    end select
    select case (trim(iname))          ! rad_trans.f90:29-37
    case ('delo'); iflag = 1
    case ('formal'); iflag = 2
    case ('lsoda', 'lsodasph')
       iflag = 0
       ! one stored point per ray: no integrator runs (grtrans_driver.f90:296-300), so the name is immaterial
       if (nup /= 1) call die('iname='//trim(iname)//': LSODA has no device implementation, use formal or delo')
    case default; call die('unknown iname '//trim(iname))
    end select

    call get_environment_variable('GRT_DEVICE', path)
    dev = 0
    if (len_trim(path) > 0) read(path, *) dev
    rc = grt_init(int(dev, c_int)); if (rc /= 0) call die_rc('grt_init', rc)
    call get_environment_variable('GRTRANS_AMD_DATA', datadir)
    if (len_trim(datadir) == 0) datadir = '.'
    path = trim(datadir)//'/polsynchpl_tables.bin'
    rc = grt_load_polsynchpl_tables_file(cstr(path)); if (rc /= 0) call die_rc('grt_load_polsynchpl_tables_file '//trim(path), rc)
    if (fluid_id == GRT_FLUID_FFJET) then
       path = trim(fdfile)              ! as load_fluid_model does (fluid.f90:226-228)
       rc = grt_load_ffjet_file(cstr(path))
       if (rc /= 0) then
          path = trim(datadir)//'/'//trim(fdfile)
          rc = grt_load_ffjet_file(cstr(path))
       end if
       if (rc /= 0) call die_rc('grt_load_ffjet_file '//trim(fdfile), rc)
    else if (fluid_id == GRT_FLUID_HARM) then
       ! initialize_harm_model(a,ifile,fdfile,fhfile,fnt,findf), fluid.f90:193-194
       ! fnt > 1: that many time slices are loaded and each point reads the one its look-back time selects (slow light)
       if (fnt > 1 .and. extra /= 0) call die('extra=1 with several HARM slices (fnt > 1) is not supported')
       rc = grt_load_harm_files_nt(cstr(trim(fdfile)), cstr(trim(fhfile)), int(findf, c_int), int(fnt, c_int), spin)
       if (rc /= 0) call die_rc('grt_load_harm_files '//trim(fdfile)//' '//trim(fhfile), rc)
       harm_dfile = fdfile; harm_hfile = fhfile; harm_indf = findf; harm_spin = spin; harm_toffset = 0.
       harm_nt = fnt
    else if (fluid_id == GRT_FLUID_HARM3D) then
       ! initialize_harm3d_model(a,ifile,fdfile,fhfile,fgfile,fnt,findf), fluid.f90:195-196; one loaded slice
       if (fnt /= 1 .or. nload /= 1) call die('HARM3D with several loaded slices (fnt, nload > 1) is not supported')
       rc = grt_load_harm3d_files(cstr(trim(fdfile)), cstr(trim(fhfile)), int(findf, c_int), spin)
       if (rc /= 0) call die_rc('grt_load_harm3d_files '//trim(fdfile)//' '//trim(fhfile), rc)
       harm_dfile = fdfile; harm_hfile = fhfile; harm_indf = findf; harm_spin = spin; harm_toffset = 0.
       harm_nt = 1
    else if (fluid_id == GRT_FLUID_KORAL3D .or. fluid_id == GRT_FLUID_KORAL3D_DISK) then
       ! initialize_koral3d_model(a,ifile,.false.,fdfile,fhfile,fnt,findf,fscalefac,...), fluid.f90:205-210; one loaded slice
       if (fnt /= 1 .or. nload /= 1) call die('KORAL3D with several loaded slices (fnt, nload > 1) is not supported')
       if (nt > 1) call die('KORAL3D movie frames (nt > 1) are not supported')
       rc = grt_load_koral3d_files(cstr(trim(fdfile)), cstr(trim(fhfile)), int(findf, c_int), spin, fscalefac)
       if (rc /= 0) call die_rc('grt_load_koral3d_files '//trim(fdfile)//' '//trim(fhfile), rc)
    else if (fluid_id == GRT_FLUID_IHARM) then
       ! initialize_iharm_model(a,ifile,fdfile,fhfile,fgfile,fnt,findf), fluid.f90:199-200; one loaded slice, metric 0
       if (fnt /= 1 .or. nload /= 1) call die('IHARM with several loaded slices (fnt, nload > 1) is not supported')
       rc = grt_load_iharm_files(cstr(trim(fdfile)), cstr(trim(fhfile)), int(findf, c_int), spin)
       if (rc /= 0) call die_rc('grt_load_iharm_files '//trim(fdfile)//' '//trim(fhfile), rc)
       harm_dfile = fdfile; harm_hfile = fhfile; harm_indf = findf; harm_spin = spin; harm_toffset = 0.
       harm_nt = 1
    else if (fluid_id == GRT_FLUID_THINDISK) then
       ! init_thindisk(real(a),ifile,real(fargs%mdot),real(fargs%mbh),real(fargs%rin),real(fargs%rout)), fluid.f90:226-227
       rc = grt_set_thindisk(fmdot, frin, frout); if (rc /= 0) call die_rc('grt_set_thindisk', rc)
    else if (fluid_id == GRT_FLUID_SPHACC) then
       ! the reference holds these tables as DATA statements (fluid_model_sphacc.f90:40-448)
       path = trim(datadir)//'/sphacc_tables.bin'
       rc = grt_load_sphacc_tables_file(cstr(path)); if (rc /= 0) call die_rc('grt_load_sphacc_tables_file '//trim(path), rc)
    end if

    write(6,*) 'welcome to grtrans! (MI355X per-pixel path)'
    allocate(ivals(nvals+nextra, npix, ncams)); ivals = 0d0
    allocate(ab(2, npix)); ab = 0d0
    ntile = i2 - i1 + 1
    allocate(pixvals(int(nvals+nextra,8)*ntile*nfreq), pixloc(2*ntile), status(ntile))
    last_ray_steps = 0d0; last_kernel_ms = 0d0
    do j = 1, nmu
       rc = grt_make_camera(cam, int(standard, c_int), gridvals, int(nn, c_int), spin, mu0(j), uout, phi0)
       if (rc /= 0) call die_rc('grt_make_camera', rc)
       if (nload > 1) then
          ! pgrtrans.f90:172-189: time origin of every ray from a one-point geodesic, normalised over the image;
          ! initialize_geo_tabs (geodesics.f90:120) leaves the mu-fill segment switched off for what follows
          cam%mufill = 0; cam%kext = 0; cam%next = 0
          allocate(t0tab(ntile))
          rc = grt_compute_t0(cam, int(i1, c_int64_t), int(i2, c_int64_t), t0tab); if (rc /= 0) call die_rc('grt_compute_t0', rc)
          where (t0tab > 0.)
             t0tab = t0tab - maxval(t0tab)
          elsewhere
             t0tab = 0.
          end where
          t0tab = t0tab - minval(t0tab)
          rc = grt_set_t0(t0tab, int(ntile, c_int64_t)); if (rc /= 0) call die_rc('grt_set_t0', rc)
          deallocate(t0tab)
       end if
       wtime = omp_get_wtime()
       do l = 1, nt
          if (l > 1) call advance_fluid_timestep(dt)             ! pgrtrans.f90:211 (after frame l-1)
          do m = 1, nmdot
             em%mbh = mbh; em%nfac = ftscl; em%bfac = frscl          ! pgrtrans.f90:125-128
             em%gmin = dble(real(gmin)); em%gmax = dble(real(gmax)); em%p1 = dble(real(p1))  ! emis.f90:32-39
             em%muval = muval; em%etype = emis_id; em%coefindx = epcoefindx
             if (nepotherargs < 2 .or. nepotherargs > 8) call die('epotherargs = (delta, 1..7 weights)')
             em%nother = nepotherargs; em%otherargs = 0d0; em%otherargs(1:nepotherargs) = epotherargs   ! pgrtrans.f90:119-120
             em%jetalpha = jetalpha; em%sigcut = sigcut               ! sparams%jetalphaval, %sigcut (pgrtrans.f90:129-136)
             em%mdot = mdots(m); em%gminval = gmin                   ! sparams%mdot, %gminval (pgrtrans.f90:127-131)
             rc = grt_trace_tile(cam, int(i1, c_int64_t), int(i2, c_int64_t), fluid_id, em, int(nfreq, c_int), freqs, &
                  int(iflag, c_int), int(nvals, c_int), int(nextra, c_int), pixvals, pixloc, status, st)
             if (rc /= 0) call die_rc('grt_trace_tile', rc)
             last_ray_steps = last_ray_steps + dble(st%ray_steps); last_kernel_ms = last_kernel_ms + st%ms_total
             do k = 1, nfreq
                ! camera index (l-1)*nfreq*nparams+(m-1)*nfreq+k with l -> (j-1)*nt+l (grtrans_driver.f90:324)
                cam_index = ((j-1)*nt + l - 1)*nfreq*nmdot + (m-1)*nfreq + k
                ivals(:, i1:i2, cam_index) = reshape(dble(pixvals((k-1)*int(nvals+nextra,8)*ntile+1 : &
                     k*int(nvals+nextra,8)*ntile)), (/nvals+nextra, ntile/))
             end do
          end do
       end do
       write(6,*) 'grtrans wall time elapsed: ', omp_get_wtime() - wtime
       write(6,*) 'ray-steps: ', last_ray_steps, ' device ms: ', last_kernel_ms
    end do
    ab(:, i1:i2) = reshape(dble(pixloc), (/2, ntile/))      ! c(1)%pixloc (pgrtrans.f90:227)
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
  end subroutine grtrans_main

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

  subroutine del_pgrtrans_data()
    if (allocated(ivals)) deallocate(ivals)
    if (allocated(ab)) deallocate(ab)
    if (allocated(freqs)) deallocate(freqs)
  end subroutine del_pgrtrans_data

  ! advance_fluid_timestep (fluid.f90:246-262) for one loaded slice: only HARM steps (advance_harm_timestep,
  ! fluid_model_harm.f90:511-521, REAL(4) tstep=20. and toffset kept across calls); the other device models are static
  subroutine advance_fluid_timestep(dtobs)
    use grt_cabi
    real(8), intent(in) :: dtobs
    integer :: nupdate, rc
    if (cur_fluid /= GRT_FLUID_HARM .and. cur_fluid /= GRT_FLUID_HARM3D .and. cur_fluid /= GRT_FLUID_IHARM) return
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
    else if (nupdate > 0 .and. cur_fluid == GRT_FLUID_HARM3D) then
       rc = grt_load_harm3d_files(cstr(trim(harm_dfile)), cstr(trim(harm_hfile)), int(harm_indf, c_int), harm_spin)
       if (rc /= 0) call die_rc('grt_load_harm3d_files (advance) '//trim(harm_dfile), rc)
    else if (nupdate > 0) then
       rc = grt_load_harm_files_nt(cstr(trim(harm_dfile)), cstr(trim(harm_hfile)), int(harm_indf, c_int), int(harm_nt, c_int), harm_spin)
       if (rc /= 0) call die_rc('grt_load_harm_files (advance) '//trim(harm_dfile), rc)
    end if
  end subroutine advance_fluid_timestep

  subroutine die(msg)
    character(len=*), intent(in) :: msg
    write(0,*) 'grtrans_amd: ', trim(msg)
    error stop 2
  end subroutine die

  subroutine die_rc(what, rc)
    character(len=*), intent(in) :: what
    integer, intent(in) :: rc
    write(0,*) 'grtrans_amd: ', trim(what), ' failed, code ', rc
    error stop 3
  end subroutine die_rc

end module pgrtrans
