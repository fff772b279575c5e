! grtrans_amd/host/grtrans.f90 -- grtrans_run(ifile,outfile): namelists -> grtrans_main (reference: grtrans.f90:2-52)
subroutine grtrans_run(ifile, outfile)
  use grtrans_inputs
  use pgrtrans
  implicit none
  character(len=*), intent(in) :: outfile, ifile
  call read_inputs(trim(ifile))
  if (i1 < 1) i1 = 1
  if (i2 < 1) i2 = nn(1)*nn(2)
  call grtrans_main(standard,mumin,mumax,nmu,phi0,spin,&
       uout,uin, rcut, nrotype, gridvals, nn,i1,i2,trim(fname), dt, nt, nload, &
       nmdot, mdotmin, mdotmax,trim(ename), mbh, nfreq, fmin, fmax, muval,&
       gmin, gmax,p1, p2, jetalpha, trim(stype),use_geokerr, nvals, trim(iname),&
       cflag, extra, debug,trim(outfile),trim(fdfile),trim(fhfile),trim(fgfile),trim(fsim),fnt,findf,fnfiles,fjonfix, &
       fnw,fnfreq_tab,fnr,foffset,fdindf,fmagcrit,frspot,fr0spot,fn0spot,ftscl,frscl, &
       fwmin,fwmax,ffmin,ffmax,frmax,fsigt,ffcol,fmdot,fnscl,fnnthscl,fnnthp,fbeta, &
       fbl06,fnp,ftp,frin,frout,fthin,fthout,fphiin,fphiout,fscalefac,sigcut,coefindx, &
       epotherargs,nepotherargs)
  call del_pgrtrans_data()
  call delete_inputs()
end subroutine grtrans_run
