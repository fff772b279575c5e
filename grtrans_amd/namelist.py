"""Fortran namelist writer with the layout grtrans' own tools produce (reference: namelist.py:1-14):
one '&name' line, one ' key=value,' line per entry, '/' terminator."""


def write_namelist(fname, names, args, vals, nargs):
    with open(fname, "w") as f:
        indx = 0
        for i, name in enumerate(names):
            f.write("&" + name + "\n")
            for _ in range(nargs[i]):
                f.write(" " + args[indx] + "=" + str(vals[indx]) + ",\n")
                indx += 1
            f.write("/\n")
