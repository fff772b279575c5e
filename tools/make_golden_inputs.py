"""Capture the exact inputs.in / files.in text the REFERENCE's grtrans_batch.py writes for the
BASELINE configs (astropy and the f2py module are stubbed: only the namelist writer runs).
Build container only; the .in files written to tests/golden/ are data."""
import os, sys, types, tempfile, shutil
sys.modules['astropy'] = types.ModuleType('astropy'); sys.modules['astropy.io'] = types.ModuleType('astropy.io')
fits = types.ModuleType('astropy.io.fits'); sys.modules['astropy.io.fits'] = fits; sys.modules['astropy.io'].fits = fits
pg = types.ModuleType('pgrtrans'); pg.pgrtrans = object(); sys.modules['pgrtrans'] = pg
sys.path.insert(0, '/root/reference')
import grtrans_batch as gr   # the reference's own module

G = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden'))
CONFIGS = {
    'ffjet': dict(fname='FFJET', jdfile='m87bl09rfp10xi5a998fluidvars.bin', nfreq=1, nmu=1, fmin=3.45e11, fmax=3.45e11,
                  ename='POLSYNCHPL', nvals=4, spin=0.998, standard=1, nn=[100, 100, 400], uout=0.01, mbh=3.4e9,
                  mumin=.906, mumax=.906, gridvals=[-40, 20, -20, 40], ntscl=2., nrscl=70.),
    'ffjet_formal25': dict(fname='FFJET', jdfile='m87bl09rfp10xi5a998fluidvars.bin', nfreq=25, nmu=1, fmin=2.41e10,
                           fmax=6.31e14, ename='POLSYNCHPL', nvals=4, spin=0.998, standard=1, nn=[1024, 1024, 400],
                           uout=0.01, mbh=3.4e9, mumin=.906, mumax=.906, gridvals=[-40, 20, -20, 40], ntscl=2.,
                           nrscl=70., iname='formal', cflag=0),
    'sphacc': dict(fname='SPHACC', nfreq=1, nmu=1, fmin=2.41e11, fmax=2.41e11, ename='POLSYNCHTH', nvals=4, spin=0.,
                   mbh=10., standard=1, uout=.04, nn=[256, 256, 200], mumin=0.5, mumax=0.5, iname='formal'),
    'thindisk': dict(),
}
d = tempfile.mkdtemp(); os.chdir(d)
for name, kw in CONFIGS.items():
    x = gr.grtrans()
    x.write_grtrans_inputs('inputs.in', **kw)
    shutil.copy('inputs.in', os.path.join(G, 'inputs_%s.in' % name))
    shutil.copy('files.in', os.path.join(G, 'files_%s.in' % name))
print('written to', G)
