"""An independent, strict reader of the FITS 4.0 container grammar (test infrastructure; shares no code with the product's
writer host/fits_min.f90 or its reader grtrans_batch.read_fits).  check(raw) walks every HDU and raises AssertionError on
any violation of: 2880-byte blocks; 80-character cards of printable ASCII; keyword field (cols 1-8: [A-Z0-9_-], left
justified, blank padded) or the HIERARCH convention; value indicator '= ' in cols 9-10; fixed-format values (logical T/F
in col 30, integers and reals right-justified ending at or before col 30 for mandatory keywords, strings quoted from col
11 with >= 8 characters between the quotes); mandatory keyword order (primary: SIMPLE, BITPIX, NAXIS, NAXISn[, EXTEND];
extension: XTENSION, BITPIX, NAXIS, NAXISn, PCOUNT, GCOUNT); END card followed by blank fill; data size from
|BITPIX| x prod NAXISn (+PCOUNT) x GCOUNT / 8, zero-filled to a block; big-endian IEEE data.  Returns
[(cards [(key, value, comment)], data ndarray)]."""
import re

import numpy as np

KEY_RE = re.compile(r"^[A-Z0-9_-]{1,8} *$")


def _value(field, mandatory):
    """field = cols 11-80.  Returns (python value, comment)."""
    if field.lstrip().startswith("'"):
        assert field.startswith("'"), "string value must start in column 11: %r" % field
        m = re.match(r"^'((?:[^']|'')*)'(.*)$", field)
        assert m, "unterminated string: %r" % field
        assert len(m.group(1)) >= 8 or not mandatory, "fixed-format string shorter than 8 characters: %r" % field
        rest = m.group(2)
        assert rest.strip() == "" or rest.lstrip().startswith("/"), "junk after string: %r" % field
        return m.group(1).replace("''", "'").rstrip(), rest.partition("/")[2].strip()
    val, _, com = field.partition("/")
    v = val.strip()
    assert v != "", "empty value field"
    if v in ("T", "F"):
        assert val[:20].rstrip().endswith(v) and len(val[:20].rstrip()) == 20, "logical not in column 30: %r" % field
        return v == "T", com.strip()
    if mandatory:
        assert len(val[:20].rstrip()) == 20 and val[:20].strip() == v, "mandatory value not right-justified in cols 11-30: %r" % field
    if re.match(r"^[+-]?\d+$", v):
        return int(v), com.strip()
    assert re.match(r"^[+-]?(\d+\.?\d*|\.\d+)([EDed][+-]?\d+)?$", v), "not a FITS number: %r" % v
    return float(v.replace("D", "E").replace("d", "e")), com.strip()


def check(raw):
    assert len(raw) % 2880 == 0 and len(raw) > 0, "file is not a whole number of 2880-byte blocks"
    pos, hdus = 0, []
    while pos < len(raw):
        cards, end = [], False
        while not end:
            block = raw[pos:pos + 2880]
            assert len(block) == 2880, "truncated header"
            pos += 2880
            for i in range(36):
                c = block[80 * i:80 * (i + 1)]
                assert all(32 <= b <= 126 for b in c), "non-printable byte in header card %r" % c
                c = c.decode("ascii")
                if end:
                    assert c == " " * 80, "non-blank card after END"
                    continue
                if c[:8] == "END     ":
                    assert c[8:] == " " * 72, "END card with content"
                    end = True
                    continue
                if c.strip() == "":
                    continue
                if c.startswith("HIERARCH"):
                    k, eq, v = c[8:].partition("=")
                    assert eq and k.strip() != "" and c[8] == " ", "bad HIERARCH card %r" % c
                    val, com = _value(v.lstrip(" "), False) if not v.lstrip().startswith("'") else _value(v.lstrip(), False)
                    cards.append((k.strip(), val, com))
                    continue
                assert KEY_RE.match(c[:8]), "bad keyword field %r" % c[:8]
                key = c[:8].rstrip()
                if key in ("COMMENT", "HISTORY") or c[8:10] != "= ":
                    cards.append((key, None, c[8:].strip()))
                    continue
                mandatory = key in ("SIMPLE", "BITPIX", "NAXIS", "EXTEND", "PCOUNT", "GCOUNT", "XTENSION") or re.match(r"^NAXIS\d+$", key)
                val, com = _value(c[10:], bool(mandatory))
                cards.append((key, val, com))
        keys = [k for k, _, _ in cards]
        first = len(hdus) == 0
        if first:
            assert keys[0] == "SIMPLE" and cards[0][1] is True, "primary HDU must start with SIMPLE = T"
        else:
            assert keys[0] == "XTENSION" and cards[0][1] == "IMAGE", "extension must start with XTENSION = 'IMAGE   '"
        assert keys[1] == "BITPIX" and cards[1][1] in (8, 16, 32, 64, -32, -64)
        assert keys[2] == "NAXIS" and 0 <= cards[2][1] <= 999
        naxis = cards[2][1]
        n = 1 if naxis else 0
        for a in range(naxis):
            assert keys[3 + a] == "NAXIS%d" % (a + 1) and cards[3 + a][1] >= 0
            n *= cards[3 + a][1]
        pcount, gcount = 0, 1
        if not first:
            assert keys[3 + naxis] == "PCOUNT" and cards[3 + naxis][1] == 0
            assert keys[4 + naxis] == "GCOUNT" and cards[4 + naxis][1] == 1
        elif len(keys) > 3 + naxis and keys[3 + naxis] == "EXTEND":
            assert cards[3 + naxis][1] in (True, False)
        assert len(set(k for k in keys if k not in ("COMMENT", "HISTORY", ""))) == len([k for k in keys if k not in ("COMMENT", "HISTORY", "")]), \
            "duplicate keyword"
        bitpix = cards[1][1]
        nbytes = abs(bitpix) // 8 * gcount * (pcount + n)
        dt = {-32: ">f4", -64: ">f8", 16: ">i2", 32: ">i4", 64: ">i8", 8: "u1"}[bitpix]
        assert pos + nbytes <= len(raw), "data unit runs past the end of the file"
        data = np.frombuffer(raw, dtype=dt, count=n, offset=pos).astype(dt[1:] if dt != "u1" else dt)
        pad = (2880 - nbytes % 2880) % 2880
        assert raw[pos + nbytes:pos + nbytes + pad] == b"\0" * pad, "data unit not zero-filled to a block"
        pos += nbytes + pad
        hdus.append((cards, data))
    return hdus
