! grtrans_amd/host/read_inputs.f90
! The inputs.in grammar of grtrans, kept verbatim: six namelists in this order
! (geodata, fluiddata, emisdata, general, harm, analytic) with the reference's variable names
! (reference: read_inputs.f90:8-20) and the derived frequency / mdot / mu0 grids (:62-98).
module grtrans_inputs
  implicit none

  integer :: standard, nrotype, nro, nphi, nup, nvals, nfreq, nmu, cflag, nt, nmdot, nload, extra, i1, i2, debug
  logical :: use_geokerr
  real(kind=8) :: mumax, mumin, spin, rcut, a1, a2, b1, b2, mbh, uout, uin, fmin, fmax, dt, mdotmin, mdotmax, &
       phi0, muval, gmin, gmax, p1, p2, jetalpha, delta, sigcut
  character(len=250) :: ename, fname, iname, stype
  real(kind=8), dimension(:), allocatable :: freqs, mdots, mu0
  real(kind=8), dimension(4) :: gridvals
  integer, dimension(3) :: nn
  character(len=250) :: fdfile, fhfile, fgfile, fsim
  integer :: fnt, findf, fnfiles, fjonfix, fnw, fnfreq_tab, fnr, foffset, fdindf, fmagcrit, fbl06, nweights, &
       nepotherargs
  integer, dimension(7) :: coefindx
  real(8) :: frspot, fr0spot, fn0spot, ftscl, frscl, fwmin, fwmax, ffmin, ffmax, frmax, fsigt, ffcol, fmdot, fnscl, &
       fnnthscl, fnnthp, fbeta, fnp, ftp, frin, frout, fthin, fthout, fphiin, fphiout, fscalefac
  real(kind=8), dimension(:), allocatable :: epotherargs

  namelist /geodata/ standard, mumin, mumax, nmu, phi0, spin, uout, uin, rcut, nrotype, gridvals, nn, i1, i2, &
       extra, debug
  namelist /fluiddata/ fname, dt, nt, nload, nmdot, mdotmin, mdotmax, sigcut
  namelist /emisdata/ ename, mbh, nfreq, fmin, fmax, muval, gmin, gmax, p1, p2, jetalpha, stype, delta, nweights, &
       coefindx
  namelist /general/ use_geokerr, nvals, iname, cflag
  namelist /harm/ fdfile, fgfile, fhfile, fnt, fnfiles, findf, fjonfix, foffset, fsim, fdindf, fmagcrit, fscalefac
  namelist /analytic/ fnw, fwmin, fwmax, fnfreq_tab, ffmin, ffmax, frmax, fnr, fsigt, ffcol, frspot, fr0spot, &
       fn0spot, ftscl, frscl, fmdot, fnscl, fnnthscl, fnnthp, fbeta, fbl06, fnp, ftp, frin, frout, fthin, fthout, &
       fphiin, fphiout

contains

  subroutine log_grid(n, lo, hi, g)
    integer, intent(in) :: n
    real(8), intent(in) :: lo, hi
    real(8), intent(out) :: g(n)
    integer :: k
    if (n == 1) then
       g(1) = lo
    else
       do k = 1, n
          g(k) = lo*exp(dble(k-1)*log(hi/lo)/dble(n-1))
       end do
    end if
  end subroutine log_grid

  subroutine read_inputs(file)
    character(len=*), intent(in) :: file
    integer :: k
    sigcut = 1d10; fscalefac = 1d0; extra = 0; debug = 0; i1 = -1; i2 = -1
    coefindx = 1; nweights = 1; delta = 1d0
    open(unit=8, file=trim(file), status='old')
    read(8, nml=geodata)
    read(8, nml=fluiddata)
    read(8, nml=emisdata)
    read(8, nml=general)
    read(8, nml=harm)
    read(8, nml=analytic)
    close(unit=8)
    a1 = gridvals(1); a2 = gridvals(2); b1 = gridvals(3); b2 = gridvals(4)
    nro = nn(1); nphi = nn(2); nup = nn(3)
    if (allocated(freqs)) deallocate(freqs)
    if (allocated(mu0)) deallocate(mu0)
    if (allocated(mdots)) deallocate(mdots)
    if (allocated(epotherargs)) deallocate(epotherargs)
    allocate(freqs(nfreq), mu0(nmu), mdots(nmdot))
    call log_grid(nfreq, fmin, fmax, freqs)
    call log_grid(nmdot, mdotmin, mdotmax, mdots)
    if (nmu == 1) then
       mu0(1) = mumin
    else
       do k = 1, nmu
          mu0(k) = mumin + (mumax-mumin)/dble(nmu-1)*dble(k-1)
       end do
    end if
    if (nweights == 1) then
       allocate(epotherargs(2))
       epotherargs(1) = delta; epotherargs(2) = 1.
    else
       allocate(epotherargs(7))
       epotherargs(1) = delta
       epotherargs(2:7) = (/0.10714295, 0.4459917, 1.34219997, 2.17450422, 1.36515333, 0.52892829/)
    end if
    nepotherargs = nweights + 1
  end subroutine read_inputs

  subroutine delete_inputs()
    if (allocated(mu0)) deallocate(mu0)
    if (allocated(freqs)) deallocate(freqs)
    if (allocated(mdots)) deallocate(mdots)
    if (allocated(epotherargs)) deallocate(epotherargs)
  end subroutine delete_inputs

end module grtrans_inputs
