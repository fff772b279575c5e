! grtrans_amd/host/grtrans_program.f90 -- ./grtrans: files.in -> grtrans_run (reference: grtrans_program.f90:1-13)
program call_grtrans
  implicit none
  interface
     subroutine grtrans_run(ifile, outfile)
       character(len=*), intent(in) :: outfile, ifile
     end subroutine grtrans_run
  end interface
  namelist /files/ ifile, ofile
  character(len=200) :: ifile, ofile
  open(unit=8, file='files.in', status='old')
  read(8, nml=files)
  close(unit=8)
  call grtrans_run(trim(ifile), trim(ofile))
end program call_grtrans
