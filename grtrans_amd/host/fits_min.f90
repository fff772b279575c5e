! grtrans_amd/host/fits_min.f90 -- a minimal FITS emitter (no cfitsio in this image).
!
! Writes what write_raytrace_camera / kwrite_raytrace_camera produce through cfitsio with cflag=1 (camera.f90:185-319):
! a primary HDU holding pixloc (BITPIX=-32, NAXIS=1, 2*nx*ny values) and one IMAGE extension per camera
! (BITPIX=-32, NAXIS=1, nx*ny*(nvals+nextra) values) whose header carries, right after the six mandatory cards,
! NX, NY, NU (the cards grtrans_batch.read_grtrans_output picks by position 6,7,8) and then the run parameters.
! FITS 4.0 fixed format: 80-character cards, 2880-byte blocks, big-endian IEEE data.  Keyword names longer than 8
! characters or with characters outside [A-Z0-9_-] use the HIERARCH convention, as cfitsio does.
module fits_min
  use iso_fortran_env, only: int8, int32, real32
  implicit none
  private
  public :: fits_open_new, fits_open_append, fits_close, fits_primary, fits_image_begin, fits_key_i, fits_key_d, fits_key_s, &
            fits_image_end
  character(len=80), allocatable :: cards(:)
  integer :: ncards = 0

contains

  subroutine fits_open_new(unit, file)
    integer, intent(in) :: unit
    character(len=*), intent(in) :: file
    open(unit=unit, file=file, access='stream', form='unformatted', status='replace', convert='big_endian')
  end subroutine fits_open_new

  subroutine fits_open_append(unit, file)
    integer, intent(in) :: unit
    character(len=*), intent(in) :: file
    open(unit=unit, file=file, access='stream', form='unformatted', status='old', position='append', convert='big_endian')
  end subroutine fits_open_append

  subroutine fits_close(unit)
    integer, intent(in) :: unit
    close(unit)
  end subroutine fits_close

  subroutine add_card(c)
    character(len=*), intent(in) :: c
    character(len=80), allocatable :: tmp(:)
    if (.not. allocated(cards)) allocate(cards(64))
    if (ncards == size(cards)) then
       allocate(tmp(2*size(cards))); tmp(1:ncards) = cards(1:ncards); call move_alloc(tmp, cards)
    end if
    ncards = ncards + 1
    cards(ncards) = c
  end subroutine add_card

  function keyname(name) result(k)
    character(len=*), intent(in) :: name
    character(len=40) :: k
    character(len=len_trim(adjustl(name))) :: t
    integer :: i, c
    logical :: plain
    t = trim(adjustl(name))
    plain = len(t) <= 8
    do i = 1, len(t)
       c = iachar(t(i:i))
       if (c >= iachar('a') .and. c <= iachar('z')) c = c - 32
       if (.not. ((c >= iachar('A') .and. c <= iachar('Z')) .or. (c >= iachar('0') .and. c <= iachar('9')) .or. &
            c == iachar('_') .or. c == iachar('-'))) plain = .false.
    end do
    if (plain) then
       k = t
       do i = 1, len(t)
          c = iachar(k(i:i))
          if (c >= iachar('a') .and. c <= iachar('z')) k(i:i) = achar(c - 32)
       end do
       k = k(1:8)//'= '
    else
       k = 'HIERARCH '//t//' = '
    end if
  end function keyname

  subroutine put(name, val, comment)
    character(len=*), intent(in) :: name, val, comment
    character(len=40) :: k
    character(len=160) :: line
    k = keyname(name)
    if (k(1:8) == 'HIERARCH') then
       line = trim(k)//' '//trim(adjustl(val))//' / '//trim(comment)
    else
       line = k(1:10)//val//' / '//trim(comment)
    end if
    call add_card(line(1:80))
  end subroutine put

  subroutine fits_key_i(name, v, comment)
    character(len=*), intent(in) :: name, comment
    integer, intent(in) :: v
    character(len=20) :: s
    write(s, '(I20)') v
    call put(name, s, comment)
  end subroutine fits_key_i

  subroutine fits_key_d(name, v, comment)
    character(len=*), intent(in) :: name, comment
    real(8), intent(in) :: v
    character(len=20) :: s
    write(s, '(ES20.12E3)') v
    call put(name, s, comment)
  end subroutine fits_key_d

  subroutine fits_key_l(name, v, comment)
    character(len=*), intent(in) :: name, comment
    logical, intent(in) :: v
    character(len=20) :: s
    s = ' '
    s(20:20) = merge('T', 'F', v)
    call put(name, s, comment)
  end subroutine fits_key_l

  subroutine fits_key_s(name, v, comment)
    character(len=*), intent(in) :: name, v, comment
    character(len=20) :: s
    character(len=18) :: body
    body = v                       ! at least 8 characters between the quotes (FITS fixed-format strings)
    s = "'"//body(1:max(8, len_trim(body)))//"'"
    call put(name, s, comment)
  end subroutine fits_key_s

  subroutine flush_header(unit)
    integer, intent(in) :: unit
    integer :: i, pad
    character(len=80) :: c
    c = 'END'
    call add_card(c)
    do i = 1, ncards
       write(unit) cards(i)
    end do
    pad = mod(36 - mod(ncards, 36), 36)
    c = ' '
    do i = 1, pad
       write(unit) c
    end do
    ncards = 0
  end subroutine flush_header

  subroutine write_data(unit, x)
    integer, intent(in) :: unit
    real(real32), intent(in) :: x(:)
    integer :: pad, i
    real(real32) :: z
    write(unit) x
    pad = mod(720 - mod(size(x), 720), 720)      ! 2880 bytes = 720 four-byte values
    z = 0.
    do i = 1, pad
       write(unit) z
    end do
  end subroutine write_data

  ! primary HDU: ftphpr(unit, simple=T, bitpix=-32, naxis=1, naxes, pcount=0, gcount=1, extend=T) + the array
  subroutine fits_primary(unit, x)
    integer, intent(in) :: unit
    real(real32), intent(in) :: x(:)
    ncards = 0
    call fits_key_l('SIMPLE', .true., 'file does conform to FITS standard')
    call fits_key_i('BITPIX', -32, 'number of bits per data pixel')
    call fits_key_i('NAXIS', 1, 'number of data axes')
    call fits_key_i('NAXIS1', size(x), 'length of data axis 1')
    call fits_key_l('EXTEND', .true., 'FITS dataset may contain extensions')
    call flush_header(unit)
    call write_data(unit, x)
  end subroutine fits_primary

  ! IMAGE extension: ftcrhd + ftphps(unit, bitpix=-32, naxis=1, naxes): the six mandatory cards; then add keys; then _end
  subroutine fits_image_begin(n)
    integer, intent(in) :: n
    ncards = 0
    call fits_key_s('XTENSION', 'IMAGE', 'IMAGE extension')
    call fits_key_i('BITPIX', -32, 'number of bits per data pixel')
    call fits_key_i('NAXIS', 1, 'number of data axes')
    call fits_key_i('NAXIS1', n, 'length of data axis 1')
    call fits_key_i('PCOUNT', 0, 'required keyword; must = 0')
    call fits_key_i('GCOUNT', 1, 'required keyword; must = 1')
  end subroutine fits_image_begin

  subroutine fits_image_end(unit, x)
    integer, intent(in) :: unit
    real(real32), intent(in) :: x(:)
    call flush_header(unit)
    call write_data(unit, x)
  end subroutine fits_image_end

end module fits_min
