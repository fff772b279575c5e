"""Python harness with the calling convention of grtrans' grtrans_batch.py, on top of the MI355X path.

Mirrors (reference grtrans_batch.py): the defaults table and kwargs (:23-170), i1/i2 clamping
(:155-164), nweights/delta (:165-170), the inputs.in / files.in writer (:172-330, :378-388),
run_grtrans (:334-357, executes ./grtrans = grtrans_amd/host/grtrans), run_pgrtrans (:397-414,
same kwargs -> pgrtrans.grtrans_main(85 positional arguments) -> ivals(nvals, npix, ncams), ab(2, npix), nu),
the binary reader (:450-477, cflag=0) and calc_spec / calc_spec_pgrtrans (:499-542).  The f2py module
`pgrtrans` of the reference is grtrans_amd/pgrtrans.py here: a ctypes view of the Fortran module pgrtrans in
host/libpgrtrans.so, the same module the executable runs (one host); there is no CPU path behind it.
"""
import os
import subprocess
import numpy as np

from . import namelist as nm
from . import capi

PKG = os.path.dirname(os.path.abspath(__file__))
HOST_EXE = os.path.join(PKG, "host", "grtrans")


class grtrans_inputs:
    def init(self):
        d = dict(cflag=1, standard=2, mumin=.1, mumax=1., phi0=-0.5, nmu=10, spin=0.998, uout=0.04, uin=1.,
                 rcut=1., nrotype=2, gridvals=[-25., 25., -25., 25.], nn=[100, 100, 1], i1=-1, i2=-1,
                 fname='THINDISK', dt=5., nt=1, nload=1, ename='BB', mbh=10., nmdot=1, mdotmin=1.5e15,
                 mdotmax=1.5e15, nfreq=1, fmin=1.e17, fmax=3.e19, muval=1. / 4., sigcut=1.e10, gmin=100,
                 gmax=1e5, p1=3.5, p2=3.5, epotherargs=[1., 1.], epcoefindx=[1, 1, 1, 1, 1, 1, 1],
                 jetalpha=0.02, stype='const', use_geokerr='T', nvals=1, iname='lsoda', tmdot=0.1, pnw=500,
                 pwmin=1.e-4, pwmax=1.e4, pnfreq_tab=100, prmax=1.e4, pnr=500, psigt=0.4, pfcol=1.7,
                 jdfile='m87bl09rfp10xi5a998fluidvars.bin', ndfile='phatdiskm8st25.bin', ntscl=1., nrscl=6.,
                 snscl=3e7, snnthscl=8e4, snnthp=2.9, sbeta=10., snp=0., snt=0., sbl06=0, srin=-1.,
                 srout=1e8, sthin=-10., sthout=10., sphiin=0., sphiout=1e4, hrspot=1.5, hr0spot=6.0,
                 hn0spot=1e4, ofile='grtrans.out', outfile='', ifile='inputs.in', hhfile='dump040',
                 hdfile='dump', hnt=1, hindf=40, tdfile='fieldline', tgfile='dump0000rr2.bin', tnt=1,
                 tnfiles=1, tindf=5206, tjonfix=1, toff=0, tsim='thickdiskrr2', tdindf=1, tmagcrit=0,
                 extra=0, debug=0, fdfile='fieldline', fgfile='dump0000rr2.bin', fhfile='dump040', fnt=1,
                 fnfiles=1, findf=5206, fjonfix=1, foffset=0, fsim='thickdiskrr2', fdindf=1, fmagcrit=0,
                 fscalefac=1.)
        self.__dict__.update(d)
        self.pfmin = self.fmin / 3.
        self.pfmax = self.fmax * 3.

    def __init__(self, **kwargs):
        self.init()
        self.__dict__.update(kwargs)
        npix = self.nn[0] * self.nn[1]
        if self.i1 < 0:
            self.i1 = 1
        if self.i2 < 0:
            self.i2 = npix
        if self.i1 > npix - 1:
            self.i1 = npix - 1
        if self.i2 > npix:
            self.i2 = npix
        self.nweights = len(self.epotherargs) - (2 if self.ename == 'MAXCOMP' else 1)
        self.nep = len(self.epotherargs)
        self.delta = self.epotherargs[0]
        # per-model overrides of the generic fluid arguments (grtrans_batch.py:219-322)
        if self.fname == 'FFJET':
            self.fdfile = self.jdfile
        elif self.fname in ('HARM', 'KORAL', 'KORALNTH', 'KORAL3D', 'KORAL3D_DISK', 'KORAL3D_TOPJET', 'KORAL3D_BOTJET'):       # (HARM3D, IHARM take the generic t* arguments, as in the reference)
            self.fdfile, self.fhfile, self.fnt, self.findf = self.hdfile, self.hhfile, self.hnt, self.hindf
        else:
            self.fdfile, self.fgfile, self.fhfile = self.tdfile, self.tgfile, self.hhfile
            self.fnt, self.fnfiles, self.findf = self.tnt, self.tnfiles, self.tindf

    def write(self, fname):
        q = lambda s: "'" + s + "'"  # noqa: E731
        csv = lambda v: ",".join(str(x) for x in v)  # noqa: E731
        names = ['geodata', 'fluiddata', 'emisdata', 'general', 'harm', 'analytic']
        blocks = [
            (['standard', 'mumin', 'mumax', 'nmu', 'phi0', 'spin', 'uout', 'uin', 'rcut', 'nrotype', 'gridvals', 'nn',
              'i1', 'i2', 'extra', 'debug'],
             [self.standard, self.mumin, self.mumax, self.nmu, self.phi0, self.spin, self.uout, self.uin, self.rcut,
              self.nrotype, csv(self.gridvals), csv(self.nn), self.i1, self.i2, self.extra, self.debug]),
            (['fname', 'dt', 'nt', 'nload', 'nmdot', 'mdotmin', 'mdotmax', 'sigcut'],
             [q(self.fname), self.dt, self.nt, self.nload, self.nmdot, self.mdotmin, self.mdotmax, self.sigcut]),
            (['ename', 'mbh', 'nfreq', 'fmin', 'fmax', 'muval', 'gmin', 'gmax', 'p1', 'p2', 'jetalpha', 'stype', 'delta',
              'nweights', 'coefindx'],
             [q(self.ename), self.mbh, self.nfreq, self.fmin, self.fmax, self.muval, self.gmin, self.gmax, self.p1,
              self.p2, self.jetalpha, q(self.stype), self.delta, self.nweights, csv(self.epcoefindx)]),
            (['use_geokerr', 'nvals', 'iname', 'cflag'], [self.use_geokerr, self.nvals, q(self.iname), self.cflag]),
            (['fdfile', 'fgfile', 'fhfile', 'fnt', 'fnfiles', 'findf', 'fjonfix', 'foffset', 'fsim', 'fdindf', 'fmagcrit',
              'fscalefac'],
             [q(self.fdfile), q(self.fgfile), q(self.fhfile), self.fnt, self.fnfiles, self.findf, self.fjonfix,
              self.foffset, q(self.fsim), self.fdindf, self.fmagcrit, self.fscalefac]),
            (['fnw', 'fwmin', 'fwmax', 'fnfreq_tab', 'ffmin', 'ffmax', 'frmax', 'fnr', 'fsigt', 'ffcol', 'frspot', 'fr0spot',
              'fn0spot', 'ftscl', 'frscl', 'fmdot', 'fnscl', 'fnnthscl', 'fnnthp', 'fbeta', 'fbl06', 'fnp', 'ftp', 'frin',
              'frout', 'fthin', 'fthout', 'fphiin', 'fphiout'],
             [self.pnw, self.pwmin, self.pwmax, self.pnfreq_tab, self.pfmin, self.pfmax, self.prmax, self.pnr, self.psigt,
              self.pfcol, self.hrspot, self.hr0spot, self.hn0spot, self.ntscl, self.nrscl, self.tmdot, self.snscl,
              self.snnthscl, self.snnthp, self.sbeta, self.sbl06, self.snp, self.snt, self.srin, self.srout, self.sthin,
              self.sthout, self.sphiin, self.sphiout])]
        args, vals, nargs = [], [], []
        for a, v in blocks:
            args += a
            vals += v
            nargs.append(len(a))
        nm.write_namelist(fname, names, args, vals, nargs)


def read_fits(path):
    """Minimal FITS reader (no astropy in this image): [(cards, data)] per HDU, cards = [(keyword, value string)],
    data = flat big-endian array converted to native float32/float64/int."""
    out = []
    with open(path, 'rb') as f:
        raw = f.read()
    pos = 0
    while pos < len(raw):
        cards, end = [], False
        while not end:
            block = raw[pos:pos + 2880]
            if len(block) < 2880:
                return out
            pos += 2880
            for i in range(36):
                c = block[80 * i:80 * (i + 1)].decode('ascii', 'replace')
                if c.startswith('END') and c[3:].strip() == '':
                    end = True
                    break
                if c.strip() == '':
                    continue
                if c.startswith('HIERARCH'):
                    k, _, v = c[8:].partition('=')
                else:
                    k, v = c[:8], c[10:] if c[8:10] == '= ' else ''
                v = v.split(' / ')[0].strip()
                cards.append((k.strip(), v.strip("' ").strip() if v.startswith("'") else v))
        hd = dict(cards)
        bitpix = int(hd.get('BITPIX', 8)); naxis = int(hd.get('NAXIS', 0))
        n = 1 if naxis else 0
        for a in range(1, naxis + 1):
            n *= int(hd['NAXIS%d' % a])
        dt = {-32: '>f4', -64: '>f8', 16: '>i2', 32: '>i4', 64: '>i8', 8: 'u1'}[bitpix]
        nbytes = n * abs(bitpix) // 8
        data = np.frombuffer(raw, dtype=dt, count=n, offset=pos).astype(dt[1:]) if n else np.zeros(0)
        pos += (nbytes + 2879) // 2880 * 2880
        out.append((cards, data))
    return out


def _rank_envs(ngpus, tile_cols, base):
    """Environment of rank r = 0..ngpus-1 of a one-process-per-GPU run (read by module pgrtrans, pgrtrans.f90)."""
    import secrets
    import tempfile
    fd, comm = tempfile.mkstemp(prefix="grtrans_amd_comm_")
    os.close(fd)
    os.remove(comm)             # rank 0 creates it (grt_comm_init_file) and removes it once every rank has joined
    nonce = secrets.token_hex(16)   # the ranks accept a rendezvous file only with this run's nonce inside
    envs = []
    for r in range(ngpus):
        e = dict(base)
        e.update(GRT_RANK=str(r), GRT_WORLD_SIZE=str(ngpus), GRT_LOCAL_RANK=str(r), GRT_COMM_FILE=comm, GRT_COMM_NONCE=nonce,
                 GRT_TILE_COLS=str(int(tile_cols)))
        e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        envs.append(e)
    return envs


def _wait_ranks(procs, grace=30.0, poll=0.2):
    """Wait for the rank processes.  RCCL calls have no timeout: when one rank dies the others would wait for it for ever, so the
    first non-zero exit starts a bounded grace period (the ranks normally leave together through the error flag their
    collectives carry) after which what is left is terminated, then killed.  Returns the exit codes."""
    import time
    t_fail = None
    while True:
        rcs = [p.poll() for p in procs]
        if all(rc is not None for rc in rcs):
            return rcs
        if t_fail is None and any(rc not in (None, 0) for rc in rcs):
            t_fail = time.time()
        if t_fail is not None and time.time() - t_fail > grace:
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            t1 = time.time()
            while time.time() - t1 < 10.0 and any(p.poll() is None for p in procs):
                time.sleep(poll)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            return [p.wait() for p in procs]
        time.sleep(poll)


def _remove_quietly(*paths):
    for q in paths:
        try:
            if q:
                os.remove(q)
        except OSError:
            pass


class grtrans:
    """run & analyze grtrans on the MI355X path (same methods as the reference's class grtrans)"""

    def write_grtrans_inputs(self, ifile, oname='grtrans.out', **kwargs):
        self.inputs = grtrans_inputs(**kwargs)
        self.inputs.write(ifile)
        self.set_grtrans_input_file(ifile, oname)

    def set_grtrans_input_file(self, fname, oname):
        nm.write_namelist('files.in', ['files'], ['ifile', 'ofile'], ['"' + fname + '"', '"' + oname + '"'], [2])
        self.ifile = fname
        self.ofile = oname

    def run_grtrans(self, exe=None, ngpus=1, tile_cols=0, **kwargs):
        """os.system('./grtrans') of the reference (grtrans_batch.py:334-357).  ngpus > 1: one ./grtrans process per GPU
        (started before any of them touches a GPU), each reads the same files.in / inputs.in, traces its share of the pixels
        and rank 0 gathers the image over RCCL and writes the output file (pgrtrans.f90 of this package)."""
        if os.path.exists(self.ofile):
            os.remove(self.ofile)
        env = dict(os.environ)
        env.setdefault("GRTRANS_AMD_DATA", capi.DATA)
        if ngpus <= 1:
            subprocess.check_call([exe or HOST_EXE], env=env)
            return
        procs = []
        envs = _rank_envs(ngpus, tile_cols, env)
        try:
            for r, e in enumerate(envs):
                procs.append(subprocess.Popen([exe or HOST_EXE], env=e, stdout=None if r == 0 else subprocess.DEVNULL))
            rcs = _wait_ranks(procs)
        finally:
            for p in procs:
                if p.poll() is None:
                    p.kill()
            _remove_quietly(envs[0]["GRT_COMM_FILE"], envs[0]["GRT_COMM_FILE"] + ".tmp")
        if any(rcs):
            raise capi.GrtError("grtrans ranks exited with %s" % rcs)

    def run_pgrtrans(self, ngpus=1, tile_cols=0, **kwargs):
        """grtrans_batch.py:397-414: same kwargs as write_grtrans_inputs; pgrtrans.grtrans_main with the 85 positional
        arguments, then copies of the module arrays ivals(nvals,npix,ncams), ab(2,npix), freqs.  `pgrtrans` is the Fortran
        module of host/libpgrtrans.so (grtrans_amd/pgrtrans.py), the same one ./grtrans runs.
        ngpus > 1: this process is rank 0; ranks 1..N-1 are child processes running the same call (grtrans_amd.rank_worker),
        started before this process touches a GPU."""
        from .pgrtrans import pgrtrans
        if len(np.shape(pgrtrans.ivals)) > 0:
            print('ERROR in run_pgrtrans: data already exist!')
            return
        self.inputs = grtrans_inputs(**kwargs)
        procs, saved, tmpfiles = [], {}, []
        watchdog_stop = None
        if ngpus > 1:
            import pickle
            import sys
            import tempfile
            import threading
            envs = _rank_envs(ngpus, tile_cols, dict(os.environ))
            job = tempfile.NamedTemporaryFile(prefix="grt_job_", suffix=".pkl", delete=False)
            pickle.dump(kwargs, job)
            job.close()
            tmpfiles = [job.name, envs[0]["GRT_COMM_FILE"], envs[0]["GRT_COMM_FILE"] + ".tmp"]
            for r in range(1, ngpus):
                procs.append(subprocess.Popen([sys.executable, "-m", "grtrans_amd.rank_worker", job.name], env=envs[r],
                                              cwd=os.getcwd(), stdout=subprocess.DEVNULL))
            for k in ("GRT_RANK", "GRT_WORLD_SIZE", "GRT_LOCAL_RANK", "GRT_COMM_FILE", "GRT_COMM_NONCE", "GRT_TILE_COLS"):
                saved[k] = os.environ.get(k)
                os.environ[k] = envs[0][k]
            # This process is rank 0 and sits inside the Fortran call: if a child rank dies without reaching the error-flag
            # exchange (a crash), rank 0 would wait in an RCCL call for ever.  The watchdog gives the orderly path a grace period,
            # then ends the job instead of hanging the session.
            watchdog_stop = threading.Event()

            def watchdog(grace=float(os.environ.get("GRT_RANK_GRACE", "60"))):
                import time
                t_fail = None
                while not watchdog_stop.wait(0.5):
                    if t_fail is None and any(p.poll() not in (None, 0) for p in procs):
                        t_fail = time.time()
                    if t_fail is not None and time.time() - t_fail > grace:
                        sys.stderr.write("grtrans_amd: a rank process died (%s) and rank 0 is still waiting for it after %.0f s: "
                                         "ending the job\n" % ([p.poll() for p in procs], grace))
                        for p in procs:
                            if p.poll() is None:
                                p.kill()
                        _remove_quietly(*tmpfiles)
                        os._exit(70)
            threading.Thread(target=watchdog, daemon=True).start()
        failed = True
        try:
            self._call_grtrans_main(pgrtrans)
            failed = False
        finally:
            if watchdog_stop is not None:
                watchdog_stop.set()
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
            if failed:          # rank 0 raised: the children are of no use any more
                for p in procs:
                    if p.poll() is None:
                        p.terminate()
            rcs = _wait_ranks(procs, grace=10.0) if procs else []
            _remove_quietly(*tmpfiles)
        if any(rcs):
            raise capi.GrtError("pgrtrans ranks exited with %s" % rcs)
        self.ivals = pgrtrans.ivals.copy()
        self.ab = pgrtrans.ab.copy()
        self.nu = pgrtrans.freqs.copy()
        self.nx = int(self.inputs.nn[0])
        self.ny = int(self.inputs.nn[1])
        self.nvals = int(np.shape(self.ivals)[0])          # nvals + nextra (the reference leaves int(inputs.nvals) here)
        self.ray_steps, self.kernel_ms = pgrtrans.last_ray_steps, pgrtrans.last_kernel_ms
        self.del_pgrtrans_data()

    def _call_grtrans_main(self, pgrtrans):
        i = self.inputs
        pgrtrans.grtrans_main(i.standard, i.mumin, i.mumax, i.nmu, i.phi0, i.spin, i.uout, i.uin, i.rcut, i.nrotype, i.gridvals, i.nn,
                              i.i1, i.i2, i.fname, i.dt, i.nt, i.nload, i.nmdot, i.mdotmin, i.mdotmax, i.ename, i.mbh, i.nfreq, i.fmin,
                              i.fmax, i.muval, i.gmin, i.gmax, i.p1, i.p2, i.jetalpha, i.stype, i.use_geokerr, i.nvals, i.iname,
                              i.cflag, i.extra, i.debug, i.outfile, i.fdfile, i.fhfile, i.fgfile, i.fsim, i.fnt, i.findf, i.fnfiles,
                              i.fjonfix, i.pnw, i.pnfreq_tab, i.pnr, i.foffset, i.fdindf, i.fmagcrit, i.hrspot, i.hr0spot, i.hn0spot,
                              i.ntscl, i.nrscl, i.pwmin, i.pwmax, i.pfmin, i.pfmax, i.prmax, i.psigt, i.pfcol, i.tmdot, i.snscl,
                              i.snnthscl, i.snnthp, i.sbeta, i.sbl06, i.snp, i.snt, i.srin, i.srout, i.sthin, i.sthout, i.sphiin,
                              i.sphiout, i.fscalefac, i.sigcut, i.epcoefindx, i.epotherargs, i.nep)

    def del_pgrtrans_data(self):
        from .pgrtrans import pgrtrans
        if len(np.shape(pgrtrans.ivals)) == 0:
            print('ERROR in del_pgrtrans_data: no data to delete!')
            return
        pgrtrans.del_pgrtrans_data()

    def read_grtrans_output(self, bin=1):
        """cflag=0 binary file (camera.f90:321-340): Fortran sequential records per camera."""
        if bin == 0:
            # FITS (cflag=1): primary HDU = pixloc, one IMAGE extension per camera; nx, ny, nu are the three cards after
            # the six mandatory ones (the reference reads them by position: hdu.header.get(6..8), grtrans_batch.py:431-434)
            hdus = read_fits(self.ofile)
            ab = hdus[0][1].reshape(-1, 2)
            nx = int(float(hdus[1][0][6][1])); ny = int(float(hdus[1][0][7][1]))
            nus = [float(h[0][8][1]) for h in hdus[1:]]
            nvals = hdus[1][1].size // (nx * ny)
            self.ab = ab
            self.ivals = np.stack([h[1].reshape(nx * ny, nvals) for h in hdus[1:]], axis=2).astype(np.float64)
            self.nu = np.array(nus)
            self.nx, self.ny, self.nvals = nx, ny, nvals
            self.calc_spec(len(nus))
            return
        ivals, nus = [], []
        with open(self.ofile, 'rb') as f:
            def rec(dtype):
                n = np.fromfile(f, dtype='i4', count=1)
                if n.size == 0:
                    return None
                a = np.fromfile(f, dtype=dtype, count=int(n[0]) // np.dtype(dtype).itemsize)
                np.fromfile(f, dtype='i4', count=1)
                return a
            while True:
                h = rec('i4')
                if h is None:
                    break
                nx, ny, nvals = [int(x) for x in h]
                nkey = int(rec('i4')[0])
                keys = [float(rec('f4')[0]) for _ in range(nkey)]
                ab = rec('f4').reshape(nx * ny, 2)
                ivals.append(rec('f4').reshape(nx * ny, nvals))
                nus.append(keys[2] if nkey >= 3 else 0.0)
        self.ab = ab
        self.ivals = np.stack(ivals, axis=2)
        self.nu = np.array(nus)
        self.nx, self.ny, self.nvals = nx, ny, nvals
        self.calc_spec(len(nus))

    def calc_freqs(self, n):
        # the run's frequency grid (grtrans_batch.py:490-497 = pgrtrans.f90:87-91)
        fmin, fmax = self.inputs.fmin, self.inputs.fmax
        self.freqs = fmin * np.exp(np.arange(n) * np.log(fmax / fmin) / (n - 1)) if n > 1 else np.array([fmin])

    def calc_spec(self, n):
        if self.ny != 1:
            da = self.ab[self.ny, 0] - self.ab[0, 0]
            db = self.ab[1, 1] - self.ab[0, 1]
            spec = np.sum(self.ivals, 0) * da * db
            if self.nvals >= 4:      # image-integrated and pixel-summed polarization fractions (grtrans_batch.py:505-509)
                with np.errstate(divide="ignore", invalid="ignore"):
                    self.lp = np.sqrt(spec[1, :] ** 2. + spec[2, :] ** 2.) / spec[0, :]
                    self.cp = spec[3, :] / spec[0, :]
                    self.lpf = np.sum(np.sqrt(self.ivals[:, 1, :] ** 2. + self.ivals[:, 2, :] ** 2.), 0) * da * db / spec[0, :]
                    self.cpf = np.sum(np.abs(self.ivals[:, 3, :]), 0) * da * db / spec[0, :]
        else:
            da = self.ab[1, 0] - self.ab[0, 0]
            db = 0.
            spec = np.array([[np.sum(self.ivals[:, j, i] * self.ab[:, 0], 0) * da * 2. * 3.14
                              for j in range(self.nvals)] for i in range(n)])
        self.spec, self.da, self.db = spec, da, db

    def calc_spec_pgrtrans(self, n):
        if self.ny != 1:
            da = self.ab[0, self.ny] - self.ab[0, 0]
            db = self.ab[1, 1] - self.ab[1, 0]
            spec = np.sum(self.ivals, 1) * da * db
            if self.nvals >= 4:      # (grtrans_batch.py:529-533)
                with np.errstate(divide="ignore", invalid="ignore"):
                    self.lp = np.sqrt(spec[1, :] ** 2. + spec[2, :] ** 2.) / spec[0, :]
                    self.cp = spec[3, :] / spec[0, :]
                    self.lpf = np.sum(np.sqrt(self.ivals[1] ** 2. + self.ivals[2] ** 2.), 0) * da * db / spec[0, :]
                    self.cpf = np.sum(np.abs(self.ivals[3]), 0) * da * db / spec[0, :]
        else:
            da = self.ab[0, 1] - self.ab[0, 0]
            db = 0.
            spec = np.array([[np.sum(self.ivals[j, :, i] * self.ab[0, :], 0) * da * 2. * np.arccos(-1.)
                              for j in range(self.nvals)] for i in range(n)])
        self.spec, self.da, self.db = spec, da, db

    # unit conversions of the file-reader layout (ivals[pixel, stokes, camera], ab[pixel, 2]); constants as the reference's harness
    # defines them (grtrans_batch.py:12-14)
    def _lbh(self):
        return 6.67259e-8 * 1.99e33 * self.inputs.mbh / 8.98755179e20

    def convert_to_lum(self):
        """spectrum and intensities to L_iso in cgs units (grtrans_batch.py:545-552)"""
        fac = 4 * np.pi * self._lbh() ** 2
        # (pixels run beta-fastest: the alpha step is ny pixels on.  The reference writes ab[self.nx,0] -- equal for its square test images)
        da = self.ab[self.ny, 0] - self.ab[0, 0]
        db = self.ab[1, 1] - self.ab[0, 1]
        self.spec = self.spec * fac
        self.ivals = self.ivals * (fac * da * db)

    def convert_to_Jy(self, D):
        """spectrum and intensities to Jy for a source at distance D in cm (grtrans_batch.py:555-562)"""
        fac = (self._lbh() ** 2 / D ** 2.) * 1e23
        da = self.ab[self.ny, 0] - self.ab[0, 0]
        db = self.ab[1, 1] - self.ab[0, 1]
        self.ivals = self.ivals * (fac * da * db)
        self.spec = self.spec * fac
