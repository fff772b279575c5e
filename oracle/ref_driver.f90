! oracle/ref_driver.f90 -- TEST INFRASTRUCTURE ONLY (never linked into the product).
!
! C-callable glue around the UNMODIFIED reference Fortran modules, which are compiled
! from /root/reference by oracle/build_ref.sh into oracle/_ref/libgrtrans_ref.so.
! Nothing here restates reference numerics: every number comes out of the reference's
! own geokerr / fluid / kerr / emissivity / radtrans_integrate modules.  The only thing
! re-stated is the *sequencing* of grtrans_driver (grtrans_driver.f90:57-466) so that every
! per-stage array of a ray can be probed.  The reference's own driver chain (grtrans_run ->
! grtrans_main -> grtrans_driver) is compiled into the same library and driven by
! oracle/refmain.py; tests/test_ref_main.py and tests/make_golden.py:crosscheck_main assert
! that the restated loop below gives the same bits (image goldens and extra=1 diagnostics).
!
! Entry points (all bind(C)):
!   ref_setup        camera + fluid model + parameter structs (pgrtrans.f90:87-171)
!   ref_camera       per-pixel camera constants (geokerr_wrapper.f:1-354)
!   ref_geokerr_ray  raw GEOKERR output arrays for one pixel (geodesics.f90:286-292)
!   ref_ray_dump     every per-step stage array of one pixel (grtrans_driver.f90:85-228)
!   ref_trace        OpenMP pixel loop i1..i2 -> pixvals, npts (pgrtrans.f90:196-212)
!   ref_teardown
!   ref_rf/rc/rd/rj, ref_sncndn, ref_polsynchpl, ref_polsynchth, ref_integrate,
!   ref_comoving_ortho, ref_ffjet_vals: leaf-level probes used to pin the C++ oracle.

module grt_ref_driver
  use iso_c_binding
  use omp_lib
  use fluid_model
  use class_rad_trans
  use geodesics
  use emissivity
  use class_four_vector
  use kerr, only: comoving_ortho, kerr_metric
  use radtrans_integrate, only: integrate, del_radtrans_integrate_data, &
       init_radtrans_integrate_data, intensity
  use polsynchemis, only: polsynchpl, polsynchth, initialize_polsynchpl, gxvals, gyvals, xvals, &
       yavals, ypvals, yvvals, yapvals, yavvals
  use phys_constants, only: GC => G, C2, MSUN
  use math, only: tsum
  implicit none

  type (geokerr_args), save :: gargs
  type (fluid_args), save :: fargs
  type (source_params), save :: sparams0
  type (emis_params), save :: eparams0
  character(len=100), save :: fname_s, ename_s, iname_s, stype_s
  integer, save :: nvals_s, npix_s, nup_s
  logical, save :: is_setup = .false.

  type (fluid) :: fl
  type (rad_trans) :: rt
  type (geo) :: ge
  type (emis) :: em
!$omp threadprivate(fl,rt,ge,em)

contains

  subroutine c2f(cstr, n, fstr)
    character(kind=c_char), dimension(*), intent(in) :: cstr
    integer(c_int), value :: n
    character(len=100), intent(out) :: fstr
    integer :: i
    fstr = ' '
    do i = 1, min(n, 100)
       if (cstr(i) == c_null_char) exit
       fstr(i:i) = cstr(i)
    end do
  end subroutine c2f

  subroutine ref_setup(standard, mu0, phi0, spin, uout, uin, rcut, nrotype, gridvals, nn, &
       fname, nfname, ename, nename, iname, niname, stype, nstype, dfile, ndfile, &
       hfile, nhfile, mbh, mdot, muval, gmin, gmax, p1, p2, jetalpha, sigcut, coefindx, &
       ftscl, frscl, fmdot, fnt, findf, nvals) bind(C, name='ref_setup')
    integer(c_int), value :: standard, nrotype, nfname, nename, niname, nstype, ndfile, nhfile, &
         fnt, findf, nvals
    real(c_double), value :: mu0, phi0, spin, uout, uin, rcut, mbh, mdot, muval, gmin, gmax, &
         p1, p2, jetalpha, sigcut, ftscl, frscl, fmdot
    real(c_double), intent(in) :: gridvals(4)
    integer(c_int), intent(in) :: nn(3), coefindx(7)
    character(kind=c_char), dimension(*), intent(in) :: fname, ename, iname, stype, dfile, hfile
    character(len=100) :: dfile_s, hfile_s, gfile_s, sim_s
    real(8) :: a
    if (is_setup) call ref_teardown()
    call c2f(fname, nfname, fname_s); call c2f(ename, nename, ename_s)
    call c2f(iname, niname, iname_s); call c2f(stype, nstype, stype_s)
    call c2f(dfile, ndfile, dfile_s); call c2f(hfile, nhfile, hfile_s)
    gfile_s = hfile_s; sim_s = 'harm'
    nvals_s = nvals; npix_s = nn(1)*nn(2); nup_s = nn(3)
    ! pgrtrans.f90:117-136
    eparams0%gmax = gmax; eparams0%p1 = p1; eparams0%p2 = p2
    if (allocated(eparams0%otherargs)) deallocate(eparams0%otherargs)
    allocate(eparams0%otherargs(2)); eparams0%otherargs = (/1d0, 1d0/)
    eparams0%coefindx = coefindx
    sparams0%mdot = mdot; sparams0%mbh = mbh
    sparams0%nfac = ftscl; sparams0%bfac = frscl
    sparams0%jetalphaval = jetalpha
    sparams0%gminval = gmin; sparams0%muval = muval
    sparams0%gmax = gmax; sparams0%p1 = p1; sparams0%p2 = p2; sparams0%sigcut = sigcut
    call assign_source_params_type(sparams0, stype_s)
    ! pgrtrans.f90:158-163 (only the arguments the in-scope models read are meaningful)
    call assign_fluid_args(fargs, dfile_s, hfile_s, gfile_s, sim_s, fnt, findf, 1, 1, &
         500, 100, 500, 0, 1, 0, 1.5d0, 6d0, 1d4, ftscl, frscl, &
         1d-4, 1d4, 3.33333333333d16, 9d19, 1d4, 0.4d0, 1.7d0, fmdot, mbh, 3d7, 8d4, 2.9d0, 10d0, &
         0, 0d0, 0d0, -1d0, 1d8, -10d0, 10d0, 0d0, 1d4, 1d0, sigcut)
    a = spin
    call load_fluid_model(fname_s, a, fargs)
    ! pgrtrans.f90:168-171,192
    call initialize_geokerr_args(gargs, npix_s)
    call initialize_pixels(gargs, .true., standard, mu0, phi0, a, uout, uin, rcut, nrotype, &
         gridvals(1), gridvals(2), gridvals(3), gridvals(4), nn(1), nn(2), nn(3))
    gargs%t0 = 0d0
    is_setup = .true.
  end subroutine ref_setup

  ! nload > 1 (pgrtrans.f90:172-189): the look-back time origin t0 of every ray from a one-point geodesic, normalised
  ! over the image.  initialize_geo_tabs also switches the mu-fill segment off for everything that follows.
  subroutine ref_nload_t0(i1, i2, t0out) bind(C, name='ref_nload_t0')
    integer(c_int), value :: i1, i2
    real(c_double), intent(out) :: t0out(npix_s)
    integer :: i
    gargs%t0 = 0d0
    gargs%nup = 1
    do i = i1, i2
       call initialize_geo_tabs(gargs, i)
    end do
    where (gargs%t0 > 0.)
       gargs%t0 = gargs%t0 - maxval(gargs%t0)
    elsewhere
       gargs%t0 = 0.
    end where
    gargs%t0 = gargs%t0 - minval(gargs%t0)
    gargs%nup = nup_s
    t0out = gargs%t0
  end subroutine ref_nload_t0

  ! advance_fluid_timestep(fname,dt) between movie frames (pgrtrans.f90:211): for HARM it re-loads dumps
  subroutine ref_advance_timestep(dt) bind(C, name='ref_advance_timestep')
    real(c_double), value :: dt
    call advance_fluid_timestep(fname_s, dt)
  end subroutine ref_advance_timestep

  ! emis_params%otherargs (pgrtrans.f90:119-120): (delta, weights...) of MAXJUTT / MAXCOMP; call after ref_setup
  subroutine ref_set_otherargs(n, args) bind(C, name='ref_set_otherargs')
    integer(c_int), value :: n
    real(c_double), intent(in) :: args(n)
    if (allocated(eparams0%otherargs)) deallocate(eparams0%otherargs)
    allocate(eparams0%otherargs(n)); eparams0%otherargs = args
  end subroutine ref_set_otherargs

  subroutine ref_teardown() bind(C, name='ref_teardown')
    if (.not. is_setup) return
    call del_geokerr_args(gargs)
    call unload_fluid_model(fname_s)
    is_setup = .false.
  end subroutine ref_teardown

  subroutine ref_camera(npix, alpha, beta, q2, l, su, sm, uf, tpr, scal) bind(C, name='ref_camera')
    integer(c_int), value :: npix
    real(c_double), intent(out) :: alpha(npix), beta(npix), q2(npix), l(npix), su(npix), &
         sm(npix), uf(npix), scal(8)
    integer(c_int), intent(out) :: tpr(npix)
    alpha = gargs%alpha(1:npix); beta = gargs%beta(1:npix); q2 = gargs%q2(1:npix)
    l = gargs%l(1:npix); su = gargs%su(1:npix); sm = gargs%sm(1:npix)
    uf = gargs%uf(1:npix); tpr = gargs%tpr(1:npix)
    scal(1) = gargs%u0; scal(2) = gargs%uout; scal(3) = gargs%offset; scal(4) = gargs%mu0
    scal(5) = gargs%a; scal(6) = gargs%kext; scal(7) = gargs%next; scal(8) = gargs%nup
  end subroutine ref_camera

  ! Raw GEOKERR arrays for pixel i (1-based), exactly the call of geodesics.f90:286-292.
  subroutine ref_geokerr_ray(i, maxn, nout, ncase, ufi, mufi, dti, dphi, lambdai, tpmi, tpri) &
       bind(C, name='ref_geokerr_ray')
    integer(c_int), value :: i, maxn
    integer(c_int), intent(out) :: nout, ncase
    real(c_double), intent(out) :: ufi(maxn), mufi(maxn), dti(maxn), dphi(maxn), lambdai(maxn)
    integer(c_int), intent(out) :: tpmi(maxn), tpri(maxn)
    real(8) :: u0, uf, uout, mu0, muf, a, l, q2, alpha, beta, su, sm, offset
    integer :: tpm, tpr, nup, phit, usegeor, mufill, kext, next, nc
    u0 = gargs%u0; uf = gargs%uf(i); uout = gargs%uout; mu0 = gargs%mu0; muf = gargs%muf(i)
    a = gargs%a; l = gargs%l(i); q2 = gargs%q2(i); alpha = gargs%alpha(i); beta = gargs%beta(i)
    tpm = gargs%tpm(i); tpr = gargs%tpr(i); su = gargs%su(i); sm = gargs%sm(i)
    nup = gargs%nup; offset = gargs%offset; phit = gargs%phit; usegeor = gargs%usegeor
    mufill = gargs%mufill; kext = gargs%kext; next = gargs%next; nc = 0
    ufi = 0d0; mufi = 0d0; dti = 0d0; dphi = 0d0; lambdai = 0d0; tpmi = 0; tpri = 0
    call geokerr(u0, uf, uout, mu0, muf, a, l, q2, alpha, beta, tpm, tpr, su, sm, nup, offset, &
         phit, usegeor, mufill, nc, kext, next, ufi, mufi, dti, dphi, tpmi, tpri, lambdai)
    nout = nup; ncase = nc
  end subroutine ref_geokerr_ray

  ! One pixel through the whole per-pixel pipeline, grtrans_driver.f90:85-228 (extra=0),
  ! for ONE frequency; dumps every stage.  dump is (maxn, 64) column-major.
  subroutine ref_ray_dump(i, freq, maxn, npts, status, fac, pix, dump) bind(C, name='ref_ray_dump')
    integer(c_int), value :: i, maxn
    real(c_double), value :: freq
    integer(c_int), intent(out) :: npts, status
    real(c_double), intent(out) :: fac, pix(4), dump(maxn, 64)
    real(8), allocatable :: s2xi(:), c2xi(:), rshift(:), ang(:), nu(:), cosne(:), tau(:)
    type (source_params) :: sp
    type (emis_params) :: ep
    real(8) :: LBH
    integer :: n, st, c
    dump = 0d0; pix = 0d0; fac = 0d0
    ep = eparams0
    call initialize_geodesic(ge, gargs, i, st)
    status = st; npts = ge%npts; n = ge%npts
    if (st == 1 .and. n > 0 .and. n <= maxn) then
       dump(1:n,1) = ge%lambda
       do c = 1, 4
          dump(1:n,1+c) = ge%x%data(c); dump(1:n,5+c) = ge%k%data(c)
       end do
       dump(1:n,10) = ge%tpmarr; dump(1:n,11) = ge%tprarr
       allocate(nu(n), s2xi(n), c2xi(n), rshift(n), ang(n), cosne(n), tau(n))
       s2xi = 0d0; c2xi = 0d0; tau = 0d0
       call initialize_fluid_model(fl, fname_s, gargs%a, n)
       call select_emissivity_values(em, ename_s)
       call get_fluid_vars(ge%x, ge%k, ge%gk%a, fl)
       dump(1:n,12) = fl%rho
       if (allocated(fl%p)) dump(1:n,13) = fl%p
       if (allocated(fl%bmag)) dump(1:n,14) = fl%bmag
       do c = 1, 4
          dump(1:n,14+c) = fl%u%data(c); dump(1:n,18+c) = fl%b%data(c)
       end do
       call comoving_ortho(ge%x%data(2), ge%x%data(3), ge%gk%a, ge%gk%alpha(1), ge%gk%beta(1), &
            ge%gk%mu0, fl%u, fl%b, ge%k, s2xi, c2xi, ang, rshift, cosne)
       dump(1:n,23) = s2xi; dump(1:n,24) = c2xi; dump(1:n,25) = ang
       dump(1:n,26) = rshift; dump(1:n,27) = cosne
       call initialize_rad_trans(rt, iname_s, n, nvals_s, 0)
       call init_radtrans_integrate_data(rt%iflag, rt%neq, n, rt%npts)
       sp = sparams0
       call initialize_source_params(sp, n)
       call initialize_emis_params(ep, n)
       call initialize_emissivity(em, n, fl%nfreq, rshift, ang, cosne, fl%nrelbin, fl%bingammamin, fl%bingammamax)
       LBH = GC*sp%mbh*MSUN/C2
       call convert_model_ref(sp)
       ep%gmin = sp%gmin; ep%mu = sp%mu
       call emis_model(em, ep)
       if (allocated(em%ncgs)) dump(1:n,28) = em%ncgs
       if (allocated(em%ncgsnth)) dump(1:n,29) = em%ncgsnth
       if (allocated(em%bcgs)) dump(1:n,30) = em%bcgs
       if (allocated(em%tcgs)) dump(1:n,31) = em%tcgs
       nu = freq/rshift
       call calc_emissivity(nu, em, ep)
       do c = 1, em%neq
          dump(1:n,31+c) = em%j(:,c)
       end do
       do c = 1, em%nk
          dump(1:n,35+c) = em%K(:,c)
       end do
       if (n > 1) then
          fac = sum(em%j(:,1)*(ge%lambda(1)-ge%lambda(1:n)))
       else
          fac = sum(em%j(:,1)/n)
       end if
       if (fac/1d16 > 0d0) then
          if (em%neq == 4) call rotate_emis(em, s2xi, c2xi)
          if (n /= 1) then
             call invariant_emis(em, rshift)
             em%j = em%j/fac; em%K = em%K*LBH
             call calc_opt_depth(ge%lambda(1:n), em, tau, 1)
             tau = -tau
             do c = 1, em%neq
                dump(1:n,42+c) = em%j(:,c)
             end do
             do c = 1, em%nk
                dump(1:n,46+c) = em%K(:,c)
             end do
             dump(1:n,54) = tau
             call integrate(ge%lambda(1:n), em%j, em%K, tau, rt%npts)
             rt%I(1:rt%neq,1:rt%npts) = intensity(1:rt%neq,1:rt%npts)
             rt%I = rt%I*fac*LBH
             do c = 1, rt%neq
                dump(1:n,54+c) = rt%I(c,1:n)
             end do
          else
             call invariant_emis(em, rshift, 3)
             rt%I(1:em%neq,1:em%npts) = transpose(em%j(1:em%npts,1:em%neq))
          end if
       else
          rt%I(:,:) = 0d0; rt%npts = 1
       end if
       pix(1:rt%neq) = dble(real(rt%I(:,rt%npts)))
       call del_source_params(sp); call del_emis_params(ep); call del_emissivity(em)
       call del_rad_trans(rt); call del_radtrans_integrate_data(); call del_fluid_model(fl)
       deallocate(nu, s2xi, c2xi, rshift, ang, cosne, tau)
    end if
    call del_geodesic(ge)
  end subroutine ref_ray_dump

  subroutine convert_model_ref(sp)   ! grtrans_driver.f90:468-481
    type (source_params), intent(inout) :: sp
    real(8), dimension(em%npts) :: ncgs, bcgs, tcgs, ncgsnth
    real(8), dimension(em%npts, fl%nrelbin) :: nnthcgs
    real(8), dimension(:), allocatable :: freqarr
    real(8), dimension(:,:), allocatable :: fnu
    call convert_fluid_vars(fl, ncgs, ncgsnth, nnthcgs, bcgs, tcgs, fnu, freqarr, sp)
    call assign_emis_params(em, ncgs, ncgsnth, nnthcgs, bcgs, tcgs, fnu, freqarr, fl%nfreq)
  end subroutine convert_model_ref

  ! One pixel, all frequencies: the body of grtrans_driver (extra=0, debug=0, nparams=1).
  subroutine trace_pixel(i, nfreq, freqs, pix, nptsout)
    integer, intent(in) :: i, nfreq
    real(8), intent(in) :: freqs(nfreq)
    real(4), intent(out) :: pix(nvals_s, nfreq)
    integer, intent(out) :: nptsout
    real(8), allocatable :: s2xi(:), c2xi(:), rshift(:), ang(:), nu(:), cosne(:), tau(:)
    type (source_params) :: sp
    type (emis_params) :: ep
    real(8) :: LBH, fac
    integer :: n, st, k
    pix = 0.
    ep = eparams0
    call initialize_geodesic(ge, gargs, i, st)
    n = ge%npts; nptsout = n
    if (st /= 1) nptsout = -n
    if (st == 1 .and. n > 0) then
       allocate(nu(n), s2xi(n), c2xi(n), rshift(n), ang(n), cosne(n), tau(n))
       s2xi = 0d0; c2xi = 0d0
       call initialize_fluid_model(fl, fname_s, gargs%a, n)
       call select_emissivity_values(em, ename_s)
       call get_fluid_vars(ge%x, ge%k, ge%gk%a, fl)
       call comoving_ortho(ge%x%data(2), ge%x%data(3), ge%gk%a, ge%gk%alpha(1), ge%gk%beta(1), &
            ge%gk%mu0, fl%u, fl%b, ge%k, s2xi, c2xi, ang, rshift, cosne)
       call initialize_rad_trans(rt, iname_s, n, nvals_s, 0)
       call init_radtrans_integrate_data(rt%iflag, rt%neq, n, rt%npts)
       sp = sparams0
       call initialize_source_params(sp, n)
       call initialize_emis_params(ep, n)
       call initialize_emissivity(em, n, fl%nfreq, rshift, ang, cosne, fl%nrelbin, fl%bingammamin, fl%bingammamax)
       LBH = GC*sp%mbh*MSUN/C2
       call convert_model_ref(sp)
       ep%gmin = sp%gmin; ep%mu = sp%mu
       call emis_model(em, ep)
       do k = 1, nfreq
          nu = freqs(k)/rshift
          call calc_emissivity(nu, em, ep)
          if (n > 1) then
             fac = sum(em%j(:,1)*(ge%lambda(1)-ge%lambda(1:n)))
          else
             fac = sum(em%j(:,1)/n)
          end if
          if (fac/1d16 > 0d0) then
             if (em%neq == 4) call rotate_emis(em, s2xi, c2xi)
             if (n /= 1) then
                call invariant_emis(em, rshift)
                em%j = em%j/fac; em%K = em%K*LBH
                call calc_opt_depth(ge%lambda(1:n), em, tau, 1)
                tau = -tau
                call integrate(ge%lambda(1:n), em%j, em%K, tau, rt%npts)
                rt%I(1:rt%neq,1:rt%npts) = intensity(1:rt%neq,1:rt%npts)
                rt%I = rt%I*fac*LBH
             else
                call invariant_emis(em, rshift, 3)
                rt%I(1:em%neq,1:em%npts) = transpose(em%j(1:em%npts,1:em%neq))
             end if
          else
             rt%I(:,:) = 0d0; rt%npts = 1
          end if
          pix(1:rt%neq,k) = real(rt%I(:,rt%npts))
       end do
       call del_source_params(sp); call del_emis_params(ep); call del_emissivity(em)
       call del_rad_trans(rt); call del_radtrans_integrate_data(); call del_fluid_model(fl)
       deallocate(nu, s2xi, c2xi, rshift, ang, cosne, tau)
    end if
    call del_geodesic(ge)
  end subroutine trace_pixel

  ! The pixel loop, pgrtrans.f90:196-212. pixvals is (nvals, npix_range, nfreq) fp32;
  ! raynpts(j) = npts of pixel i1+j-1 (0 = never reached uout, negative = status -1).
  subroutine ref_trace(i1, i2, nfreq, freqs, nthreads, pixvals, raynpts, seconds) bind(C, name='ref_trace')
    integer(c_int), value :: i1, i2, nfreq, nthreads
    real(c_double), intent(in) :: freqs(nfreq)
    real(c_float), intent(out) :: pixvals(nvals_s, i2-i1+1, nfreq)
    integer(c_int), intent(out) :: raynpts(i2-i1+1)
    real(c_double), intent(out) :: seconds
    real(4) :: pix(nvals_s, nfreq)
    integer :: i, np
    real(8) :: t0
    if (nthreads > 0) call omp_set_num_threads(nthreads)
    t0 = omp_get_wtime()
!$omp parallel do schedule(static,1) private(i,pix,np)
    do i = i1, i2
       call trace_pixel(i, nfreq, freqs, pix, np)
       pixvals(:, i-i1+1, :) = pix
       raynpts(i-i1+1) = np
    end do
!$omp end parallel do
    seconds = omp_get_wtime() - t0
  end subroutine ref_trace

  ! One pixel, one frequency, EXTRA_QUANTS=1 and npts>1: the 19 ray diagnostics of grtrans_driver.f90:230-292
  ! appended to the Stokes values.  Sequencing restated (grtrans_driver.f90 cannot be compiled here, see header);
  ! every cumulative integral is the reference's own calc_opt_depth / tsum.
  subroutine trace_pixel_extra(i, freq, pix, nptsout)
    integer, intent(in) :: i
    real(8), intent(in) :: freq
    real(4), intent(out) :: pix(nvals_s + 19)
    integer, intent(out) :: nptsout
    real(8), allocatable :: s2xi(:), c2xi(:), rshift(:), ang(:), nu(:), cosne(:), tau(:)
    real(8), allocatable :: tau_arr(:,:), w(:), q(:), tq(:), tw(:), dlp(:), lp(:)
    type (source_params) :: sp
    type (emis_params) :: ep
    real(8) :: LBH, fac, xt(19), sd
    integer :: n, st, ii, taudex, m
    integer, parameter :: cols(6) = (/1, 2, 3, 4, 5, 7/)
    pix = 0.
    ep = eparams0
    call initialize_geodesic(ge, gargs, i, st)
    n = ge%npts; nptsout = n
    if (st /= 1) nptsout = -n
    if (st == 1 .and. n > 1) then
       allocate(nu(n), s2xi(n), c2xi(n), rshift(n), ang(n), cosne(n), tau(n))
       s2xi = 0d0; c2xi = 0d0
       call initialize_fluid_model(fl, fname_s, gargs%a, n)
       call select_emissivity_values(em, ename_s)
       call get_fluid_vars(ge%x, ge%k, ge%gk%a, fl)
       call comoving_ortho(ge%x%data(2), ge%x%data(3), ge%gk%a, ge%gk%alpha(1), ge%gk%beta(1), &
            ge%gk%mu0, fl%u, fl%b, ge%k, s2xi, c2xi, ang, rshift, cosne)
       call initialize_rad_trans(rt, iname_s, n, nvals_s, 1)
       call init_radtrans_integrate_data(rt%iflag, rt%neq, n, rt%npts)
       sp = sparams0
       call initialize_source_params(sp, n)
       call initialize_emis_params(ep, n)
       call initialize_emissivity(em, n, fl%nfreq, rshift, ang, cosne, fl%nrelbin, fl%bingammamin, fl%bingammamax)
       LBH = GC*sp%mbh*MSUN/C2
       call convert_model_ref(sp)
       ep%gmin = sp%gmin; ep%mu = sp%mu
       call emis_model(em, ep)
       nu = freq/rshift
       call calc_emissivity(nu, em, ep)
       fac = sum(em%j(:,1)*(ge%lambda(1)-ge%lambda(1:n)))
       xt = 0d0
       if (fac/1d16 > 0d0) then
          if (em%neq == 4) call rotate_emis(em, s2xi, c2xi)
          call invariant_emis(em, rshift)
          em%j = em%j/fac; em%K = em%K*LBH
          call calc_opt_depth(ge%lambda(1:n), em, tau, 1)
          tau = -tau
          call integrate(ge%lambda(1:n), em%j, em%K, tau, rt%npts)
          rt%I(1:rt%neq,1:rt%npts) = intensity(1:rt%neq,1:rt%npts)
          rt%I = rt%I*fac*LBH
          ! --- :230-292 ---
          allocate(tau_arr(n,6), w(n), q(n), tq(n), tw(n), dlp(n-1), lp(n))
          do ii = 1, 6
             call calc_opt_depth(ge%lambda(1)-ge%lambda(1:n), em, tq, cols(ii))
             tau_arr(:,ii) = tq
          end do
          taudex = minloc(abs(tau-1d0), 1)
          if (minval(tau) < 1d0) taudex = n
          if (taudex == 1) taudex = 2
          xt(1:6) = tau_arr(taudex,:)
          w = em%j(:,1)*exp(-tau(1:n))
          tw = tsum(ge%lambda, w)
          do m = 7, 14
             select case (m)
             case (7);  q = ge%x%data(2)
             case (8);  q = ge%x%data(3)
             case (9);  q = ge%x%data(4)
             case (10); q = em%ncgs
             case (11); q = em%tcgs
             case (12); q = em%bcgs
             case (13); q = fl%p*2./fl%bmag**2.
             case (14); q = sign(1d0, cos(ge%x%data(3)))
             end select
             tq = tsum(ge%lambda, w*q)
             xt(m) = 1d0/tw(taudex)*tq(taudex)
          end do
          if (rt%neq == 4) then
             do m = 1, n      ! rt%I is stored in integration order: column n+1-m is stored point m
                lp(m) = sqrt(rt%I(1,n+1-m)**2. + rt%I(2,n+1-m)**2.)
             end do
             dlp = lp(1:n-1) - lp(2:n)
          else
             dlp = 0.
          end if
          sd = sum(dlp)
          xt(15) = sum(dlp*ge%x(1:n-1)%data(2))/sd
          xt(16) = sum(dlp*ge%x(1:n-1)%data(3))/sd
          xt(17) = sum(dlp*tau_arr(1:n-1,5))/sd
          xt(18) = sum(dlp*tau_arr(1:n-1,6))/sd
          xt(19) = sum(dlp*sign(1d0, cos(ge%x(1:n-1)%data(3))))/sd
          deallocate(tau_arr, w, q, tq, tw, dlp, lp)
       else
          rt%I(:,:) = 0d0; rt%npts = 1
       end if
       pix(1:rt%neq) = real(rt%I(:,rt%npts))
       pix(nvals_s+1:nvals_s+19) = real(xt)
       call del_source_params(sp); call del_emis_params(ep); call del_emissivity(em)
       call del_rad_trans(rt); call del_radtrans_integrate_data(); call del_fluid_model(fl)
       deallocate(nu, s2xi, c2xi, rshift, ang, cosne, tau)
    end if
    call del_geodesic(ge)
  end subroutine trace_pixel_extra

  subroutine ref_trace_extra(i1, i2, freq, nthreads, pixvals, raynpts) bind(C, name='ref_trace_extra')
    integer(c_int), value :: i1, i2, nthreads
    real(c_double), value :: freq
    real(c_float), intent(out) :: pixvals(nvals_s + 19, i2-i1+1)
    integer(c_int), intent(out) :: raynpts(i2-i1+1)
    real(4) :: pix(nvals_s + 19)
    integer :: i, np
    if (nthreads > 0) call omp_set_num_threads(nthreads)
!$omp parallel do schedule(static,1) private(i,pix,np)
    do i = i1, i2
       call trace_pixel_extra(i, freq, pix, np)
       pixvals(:, i-i1+1) = pix
       raynpts(i-i1+1) = np
    end do
!$omp end parallel do
  end subroutine ref_trace_extra

  ! ---------------- leaf probes ----------------
  function ref_rf(x, y, z) bind(C, name='ref_rf') result(v)
    real(c_double), value :: x, y, z
    real(c_double) :: v
    real(8), external :: rf
    v = rf(x, y, z)
  end function ref_rf
  function ref_rc(x, y) bind(C, name='ref_rc') result(v)
    real(c_double), value :: x, y
    real(c_double) :: v
    real(8), external :: rc
    v = rc(x, y)
  end function ref_rc
  function ref_rd(x, y, z) bind(C, name='ref_rd') result(v)
    real(c_double), value :: x, y, z
    real(c_double) :: v
    real(8), external :: rd
    v = rd(x, y, z)
  end function ref_rd
  function ref_rj(x, y, z, p) bind(C, name='ref_rj') result(v)
    real(c_double), value :: x, y, z, p
    real(c_double) :: v
    real(8), external :: rj
    v = rj(x, y, z, p)
  end function ref_rj
  subroutine ref_sncndn(uu, emmc, sn, cn, dn) bind(C, name='ref_sncndn')
    real(c_double), value :: uu, emmc
    real(c_double), intent(out) :: sn, cn, dn
    call sncndn(uu, emmc, sn, cn, dn)
  end subroutine ref_sncndn

  subroutine ref_polsynchpl(n, nu, nnth, b, th, p, gmin, gmax, eout) bind(C, name='ref_polsynchpl')
    integer(c_int), value :: n
    real(c_double), value :: gmax
    real(c_double), intent(in) :: nu(n), nnth(n), b(n), th(n), p(n), gmin(n)
    real(c_double), intent(out) :: eout(n, 11)
    call initialize_polsynchpl(4)
    call polsynchpl(nu, nnth, b, th, p, gmin, gmax, eout)
  end subroutine ref_polsynchpl

  subroutine ref_polsynchth(n, nu, dens, b, t, th, eout) bind(C, name='ref_polsynchth')
    integer(c_int), value :: n
    real(c_double), intent(in) :: nu(n), dens(n), b(n), t(n), th(n)
    real(c_double), intent(out) :: eout(n, 11)
    call polsynchth(nu, dens, b, t, th, eout)
  end subroutine ref_polsynchth

  ! integrate() on caller-provided coefficients: iflag 1 delo, 2 formal (0 lsoda).
  subroutine ref_integrate(iflag, n, x, jin, Kin, tauin, iout) bind(C, name='ref_integrate')
    integer(c_int), value :: iflag, n
    real(c_double), intent(in) :: x(n), jin(n,4), Kin(n,7), tauin(n)
    real(c_double), intent(out) :: iout(4, n)
    integer :: rn
    rn = n
    call init_radtrans_integrate_data(iflag, 4, n, n)
    call integrate(x, jin, Kin, tauin, rn)
    iout = intensity(1:4, 1:n)
    call del_radtrans_integrate_data()
  end subroutine ref_integrate

  subroutine ref_comoving_ortho(n, rr, th, a, alpha, beta, mus, u, b, k, out) bind(C, name='ref_comoving_ortho')
    integer(c_int), value :: n
    real(c_double), value :: a, alpha, beta, mus
    real(c_double), intent(in) :: rr(n), th(n), u(n,4), b(n,4), k(n,4)
    real(c_double), intent(out) :: out(n, 5)
    type (four_vector), dimension(n) :: uu, bb, kk
    real(8), dimension(n) :: s2xi, c2xi, ang, gg, cosne
    integer :: c
    do c = 1, 4
       uu%data(c) = u(:,c); bb%data(c) = b(:,c); kk%data(c) = k(:,c)
    end do
    call assign_metric(uu, transpose(kerr_metric(rr, th, a)))
    call assign_metric(bb, transpose(kerr_metric(rr, th, a)))
    call assign_metric(kk, transpose(kerr_metric(rr, th, a)))
    call comoving_ortho(rr, th, a, alpha, beta, mus, uu, bb, kk, s2xi, c2xi, ang, gg, cosne)
    out(:,1) = s2xi; out(:,2) = c2xi; out(:,3) = ang; out(:,4) = gg; out(:,5) = cosne
  end subroutine ref_comoving_ortho

  ! The tabulated log G(x) integrals of polsynchpl as the compiled reference holds them
  ! (polsynchemis.f90:107-525): out = gxvals(181) gyvals(543) xvals(81) yavals(243)
  ! ypvals(243) yvvals(243) yapvals(243) yavvals(243) = 2020 doubles.
  subroutine ref_polsynchpl_tables(out) bind(C, name='ref_polsynchpl_tables')
    real(c_double), intent(out) :: out(2020)
    call initialize_polsynchpl(4)
    out(1:181) = gxvals; out(182:724) = gyvals; out(725:805) = xvals; out(806:1048) = yavals
    out(1049:1291) = ypvals; out(1292:1534) = yvvals; out(1535:1777) = yapvals
    out(1778:2020) = yavvals
  end subroutine ref_polsynchpl_tables

  ! The SPHACC 1-D tables as the compiled reference holds them (fluid_model_sphacc.f90:13-443):
  ! out = uvals(498) vvals(498) rvals(461) tvals(461), REAL(4)
  subroutine ref_sphacc_tables(out) bind(C, name='ref_sphacc_tables')
    use fluid_model_sphacc, only: uvals, vvals, rvals, tvals, init_sphacc, del_sphacc
    real(c_float), intent(out) :: out(1918)
    logical :: mine
    mine = .not. allocated(uvals)
    if (mine) call init_sphacc()
    out(1:498) = uvals; out(499:996) = vvals; out(997:1457) = rvals; out(1458:1918) = tvals
    if (mine) call del_sphacc()
  end subroutine ref_sphacc_tables

end module grt_ref_driver
