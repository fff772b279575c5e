! grtrans_amd/host/grt_cabi.f90
! ISO_C_BINDING view of include/grtrans_amd.h: the only door between the Fortran 90 host
! (namelists, parameter grids, result arrays, output files) and the gfx950 kernels.
module grt_cabi
  use iso_c_binding
  implicit none

  integer(c_int), parameter :: GRT_FLUID_FFJET = 13, GRT_FLUID_HARM = 12, GRT_FLUID_SPHACC = 1, &
       GRT_FLUID_THINDISK = 2, GRT_FLUID_HARM3D = 19, GRT_FLUID_IHARM = 29, &
       GRT_FLUID_KORAL3D = 25, GRT_FLUID_KORAL3D_DISK = 26
  integer(c_int), parameter :: GRT_EMIS_POLSYNCHTH = 2, GRT_EMIS_SYNCHPL = 3, GRT_EMIS_POLSYNCHPL = 4, &
       GRT_EMIS_BB = 8, GRT_EMIS_BBPOL = 9, GRT_EMIS_HYBRIDTHPL = 20, GRT_EMIS_HYBRIDTH = 21, GRT_EMIS_HYBRIDPL = 22, &
       GRT_EMIS_MAXJUTT = 23, GRT_EMIS_MAXCOMP = 24, GRT_EMIS_SYNCHTHAV = 13, GRT_EMIS_SYNCHTHAVNOABS = 14, &
       GRT_EMIS_BREMS = 5, GRT_EMIS_SYNCHTHBREMS = 27, GRT_EMIS_RHO = 12, GRT_EMIS_POLSYNCHSYMTH = 15

  type, bind(C) :: grt_camera          ! GrtCam
     real(c_double) :: a1, a2, b1, b2
     real(c_double) :: a, mu0, uout, u0, offset, phi0
     integer(c_int32_t) :: standard, nro, nphi, nup
     integer(c_int32_t) :: usegeor, phit, mufill, kext, next, nrotype
     real(c_double) :: rcut
  end type grt_camera

  type, bind(C) :: grt_emis            ! GrtEmis
     real(c_double) :: mbh, nfac, bfac
     real(c_double) :: gmin, gmax, p1
     real(c_double) :: muval
     real(c_double) :: mdot, gminval
     real(c_double) :: jetalpha, sigcut
     integer(c_int32_t) :: etype
     integer(c_int32_t) :: coefindx(7)
     integer(c_int32_t) :: nother, pad1
     real(c_double) :: otherargs(8)
  end type grt_emis

  type, bind(C) :: grt_stats           ! GrtStats
     integer(c_int64_t) :: ray_steps, rays_live, rays_dead
     real(c_double) :: ms_setup, ms_trace, ms_total, ms_geo, ms_transfer
  end type grt_stats

  interface
     function grt_init(device) bind(C, name='grt_init') result(rc)
       import :: c_int
       integer(c_int), value :: device
       integer(c_int) :: rc
     end function grt_init
     function grt_free() bind(C, name='grt_free') result(rc)
       import :: c_int
       integer(c_int) :: rc
     end function grt_free
     function grt_load_ffjet_file(path) bind(C, name='grt_load_ffjet_file') result(rc)
       import :: c_int, c_char
       character(kind=c_char), dimension(*), intent(in) :: path
       integer(c_int) :: rc
     end function grt_load_ffjet_file
     function grt_load_harm_files(dfile, hfile, indf, spin) bind(C, name='grt_load_harm_files') result(rc)
       import :: c_int, c_char, c_double
       character(kind=c_char), dimension(*), intent(in) :: dfile, hfile
       integer(c_int), value :: indf
       real(c_double), value :: spin
       integer(c_int) :: rc
     end function grt_load_harm_files
     function grt_compute_t0(cam, i1, i2, t0) bind(C, name='grt_compute_t0') result(rc)
       import :: c_int, c_int64_t, c_double, grt_camera
       type(grt_camera), intent(in) :: cam
       integer(c_int64_t), value :: i1, i2
       real(c_double), intent(out) :: t0(*)
       integer(c_int) :: rc
     end function grt_compute_t0
     function grt_set_t0(t0, n) bind(C, name='grt_set_t0') result(rc)
       import :: c_int, c_int64_t, c_double
       real(c_double), intent(in) :: t0(*)
       integer(c_int64_t), value :: n
       integer(c_int) :: rc
     end function grt_set_t0
     function grt_set_thindisk(mdot, rin, rout) bind(C, name='grt_set_thindisk') result(rc)
       import :: c_int, c_double
       real(c_double), value :: mdot, rin, rout
       integer(c_int) :: rc
     end function grt_set_thindisk
     function grt_load_koral3d_files(dfile, hfile, indf, spin, scalefac) bind(C, name='grt_load_koral3d_files') result(rc)
       import :: c_char, c_int, c_double
       character(kind=c_char), dimension(*), intent(in) :: dfile, hfile
       integer(c_int), value :: indf
       real(c_double), value :: spin, scalefac
       integer(c_int) :: rc
     end function grt_load_koral3d_files
     function grt_load_koral3d_files_nt(dfile, hfile, indf, nt, spin, scalefac) bind(C, name='grt_load_koral3d_files_nt') result(rc)
       import :: c_char, c_int, c_double
       character(kind=c_char), dimension(*), intent(in) :: dfile, hfile
       integer(c_int), value :: indf, nt
       real(c_double), value :: spin, scalefac
       integer(c_int) :: rc
     end function grt_load_koral3d_files_nt
     function grt_get_koral3d_tstep(tstep) bind(C, name='grt_get_koral3d_tstep') result(rc)
       import :: c_int, c_float
       real(c_float), intent(out) :: tstep
       integer(c_int) :: rc
     end function grt_get_koral3d_tstep
     function grt_load_iharm_files(dfile, hfile, indf, spin) bind(C, name='grt_load_iharm_files') result(rc)
       import :: c_char, c_int, c_double
       character(kind=c_char), dimension(*), intent(in) :: dfile, hfile
       integer(c_int), value :: indf
       real(c_double), value :: spin
       integer(c_int) :: rc
     end function grt_load_iharm_files
     function grt_load_harm3d_files(dfile, hfile, indf, spin) bind(C, name='grt_load_harm3d_files') result(rc)
       import :: c_char, c_int, c_double
       character(kind=c_char), dimension(*), intent(in) :: dfile, hfile
       integer(c_int), value :: indf
       real(c_double), value :: spin
       integer(c_int) :: rc
     end function grt_load_harm3d_files
     function grt_load_harm_files_nt(dfile, hfile, indf, nt, spin) bind(C, name='grt_load_harm_files_nt') result(rc)
       import :: c_int, c_char, c_double
       character(kind=c_char), dimension(*), intent(in) :: dfile, hfile
       integer(c_int), value :: indf, nt
       real(c_double), value :: spin
       integer(c_int) :: rc
     end function grt_load_harm_files_nt
     function grt_comm_init_file(rank, nranks, path, timeout_s) bind(C, name='grt_comm_init_file') result(rc)
       import :: c_int, c_char, c_double
       integer(c_int), value :: rank, nranks
       character(kind=c_char), dimension(*), intent(in) :: path
       real(c_double), value :: timeout_s
       integer(c_int) :: rc
     end function grt_comm_init_file
     function grt_comm_free() bind(C, name='grt_comm_free') result(rc)
       import :: c_int
       integer(c_int) :: rc
     end function grt_comm_free
     function grt_comm_allreduce(vals, n, op) bind(C, name='grt_comm_allreduce') result(rc)
       import :: c_int, c_double
       real(c_double), intent(inout) :: vals(*)
       integer(c_int), value :: n, op
       integer(c_int) :: rc
     end function grt_comm_allreduce
     function grt_stage_reset(ncams, nvals, npix_rank) bind(C, name='grt_stage_reset') result(rc)
       import :: c_int, c_int64_t
       integer(c_int), value :: ncams, nvals
       integer(c_int64_t), value :: npix_rank
       integer(c_int) :: rc
     end function grt_stage_reset
     function grt_stage_resident(cam0, nfreq, npix_tile, pix_offset) bind(C, name='grt_stage_resident') result(rc)
       import :: c_int, c_int64_t
       integer(c_int), value :: cam0, nfreq
       integer(c_int64_t), value :: npix_tile, pix_offset
       integer(c_int) :: rc
     end function grt_stage_resident
     function grt_gather_stage(npix_of_rank, root, host_full) bind(C, name='grt_gather_stage') result(rc)
       import :: c_int, c_int64_t, c_float
       integer(c_int64_t), intent(in) :: npix_of_rank(*)
       integer(c_int), value :: root
       real(c_float), intent(out) :: host_full(*)
       integer(c_int) :: rc
     end function grt_gather_stage
     function grt_camera_pixels(cam, i1, i2, alpha, beta, q2, l, su, sm, uf, muf, tpm, tpr) bind(C, name='grt_camera_pixels') result(rc)
       import :: c_int, c_int64_t, c_double, c_ptr, grt_camera
       type(grt_camera), intent(in) :: cam
       integer(c_int64_t), value :: i1, i2
       real(c_double), intent(out) :: alpha(*), beta(*)
       type(c_ptr), value :: q2, l, su, sm, uf, muf, tpm, tpr      ! optional outputs: c_null_ptr to skip
       integer(c_int) :: rc
     end function grt_camera_pixels
     function grt_get_harm_tstep(tstep) bind(C, name='grt_get_harm_tstep') result(rc)
       import :: c_int, c_float
       real(c_float), intent(out) :: tstep
       integer(c_int) :: rc
     end function grt_get_harm_tstep
     function grt_load_chandra_file(path) bind(C, name='grt_load_chandra_file') result(rc)
       import :: c_int, c_char
       character(kind=c_char), dimension(*), intent(in) :: path
       integer(c_int) :: rc
     end function grt_load_chandra_file
     function grt_load_sphacc_tables_file(path) bind(C, name='grt_load_sphacc_tables_file') result(rc)
       import :: c_int, c_char
       character(kind=c_char), dimension(*), intent(in) :: path
       integer(c_int) :: rc
     end function grt_load_sphacc_tables_file
     function grt_load_polsynchpl_tables_file(path) bind(C, name='grt_load_polsynchpl_tables_file') result(rc)
       import :: c_int, c_char
       character(kind=c_char), dimension(*), intent(in) :: path
       integer(c_int) :: rc
     end function grt_load_polsynchpl_tables_file
     function grt_make_camera(cam, standard, gridvals, nn, spin, mu0, uout, phi0) &
          bind(C, name='grt_make_camera') result(rc)
       import :: c_int, c_double, grt_camera
       type(grt_camera), intent(out) :: cam
       integer(c_int), value :: standard
       real(c_double), intent(in) :: gridvals(4)
       integer(c_int), intent(in) :: nn(3)
       real(c_double), value :: spin, mu0, uout, phi0
       integer(c_int) :: rc
     end function grt_make_camera
     function grt_set_camera_polar(cam, rcut) bind(C, name='grt_set_camera_polar') result(rc)
       import :: c_int, c_double, grt_camera
       type(grt_camera), intent(inout) :: cam
       real(c_double), value :: rcut
       integer(c_int) :: rc
     end function grt_set_camera_polar
     function grt_trace_tile(cam, i1, i2, fluid_model, em, nfreq, freqs, iflag, nvals, nextra, pixvals, pixloc, &
          ray_status, st) bind(C, name='grt_trace_tile') result(rc)
       import :: c_int, c_int64_t, c_double, c_float, c_int32_t, grt_camera, grt_emis, grt_stats
       type(grt_camera), intent(in) :: cam
       integer(c_int64_t), value :: i1, i2
       integer(c_int), value :: fluid_model, nfreq, iflag, nvals, nextra
       type(grt_emis), intent(in) :: em
       real(c_double), intent(in) :: freqs(*)
       real(c_float), intent(out) :: pixvals(*), pixloc(*)
       integer(c_int32_t), intent(out) :: ray_status(*)
       type(grt_stats), intent(out) :: st
       integer(c_int) :: rc
     end function grt_trace_tile
     ! debug=1: the arrays of geodebug.out for one pixel and frequency (grtrans_driver.f90:94-110, 341-421)
     function grt_debug_ray(cam, pixel, fluid_model, em, freq, iflag, nvals, maxpts, npts, header, rows, fac) &
          bind(C, name='grt_debug_ray') result(rc)
       import :: c_int, c_int64_t, c_double, c_int32_t, grt_camera, grt_emis
       type(grt_camera), intent(in) :: cam
       integer(c_int64_t), value :: pixel
       integer(c_int), value :: fluid_model, iflag, nvals, maxpts
       type(grt_emis), intent(in) :: em
       real(c_double), value :: freq
       integer(c_int32_t), intent(out) :: npts
       real(c_double), intent(out) :: header(11), rows(*), fac
       integer(c_int) :: rc
     end function grt_debug_ray
     ! the same loop over several pixel ranges in one set of launches (a rank's column tiles); output rows follow the ranges
     function grt_trace_ranges(cam, nranges, lo, hi, fluid_model, em, nfreq, freqs, iflag, nvals, nextra, pixvals, pixloc, &
          ray_status, st) bind(C, name='grt_trace_ranges') result(rc)
       import :: c_int, c_int64_t, c_double, c_float, c_int32_t, grt_camera, grt_emis, grt_stats
       type(grt_camera), intent(in) :: cam
       integer(c_int), value :: nranges
       integer(c_int64_t), intent(in) :: lo(*), hi(*)
       integer(c_int), value :: fluid_model, nfreq, iflag, nvals, nextra
       type(grt_emis), intent(in) :: em
       real(c_double), intent(in) :: freqs(*)
       real(c_float), intent(out) :: pixvals(*), pixloc(*)
       integer(c_int32_t), intent(out) :: ray_status(*)
       type(grt_stats), intent(out) :: st
       integer(c_int) :: rc
     end function grt_trace_ranges
  end interface

contains

  function cstr(s) result(c)
    character(len=*), intent(in) :: s
    character(kind=c_char), dimension(len_trim(s)+1) :: c
    integer :: i
    do i = 1, len_trim(s)
       c(i) = s(i:i)
    end do
    c(len_trim(s)+1) = c_null_char
  end function cstr

end module grt_cabi
