"""Scan AMDGPU assembly (hipcc -S, or llvm-objdump -d of the code object) for vector copies / spills the register allocator placed in
the FLOW block of a divergent if-else BEFORE the instruction that flips the exec mask (s_andn2_saveexec / s_or_saveexec ... s_xor exec):
such a copy runs for the lanes of the `then` side only, and the lanes of the `else` side later read a stale register.  r03 found
`facs` of grt_transfer_kernel<FORMAL, IHARM, HYBRIDTHPL, 1, false> restored from another variable's register that way
(DESIGN.md 6).  r04: the same placement of a SPILL -- scratch_store of a register that lives through the if-else, before the flip,
reloaded from the same slot later -- and of v_accvgpr_write / v_accvgpr_read parks is flagged too.  SGPR spills are NOT a hazard of this
kind: v_writelane_b32 / v_readlane_b32 ignore the exec mask (they address one lane by number), so a lane written under a partial mask is
written all the same; the scanner counts them per flow block only to report that they were looked at.
Usage: python tools/check_flow_copies.py file.s [more.s]  -> one line per suspicious site, exit 1 if any."""
import re
import sys

VEC_WRITE = re.compile(r"^\s*(v_mov_b(32|64)|v_accvgpr_(write|read)|v_pk_mov|scratch_store|scratch_load|buffer_store|buffer_load)")
EXEC_FLIP = re.compile(r"^\s*(s_andn2_saveexec_b64|s_or_saveexec_b64|s_xor_saveexec_b64)")
EXEC_MASKED_ENTRY = re.compile(r"^\s*s_and_saveexec_b64")
LABEL = re.compile(r"^(\.LBB\d+_\d+:|; %bb\.\d+:)")
FUNC = re.compile(r"^(_Z\w+|\w+):\s*(;.*)?$")


def regset(tok):
    m = re.match(r"([av])\[(\d+):(\d+)\]", tok)
    if m:
        return set((m.group(1), r) for r in range(int(m.group(2)), int(m.group(3)) + 1))
    m = re.match(r"([av])(\d+)", tok)
    return set([(m.group(1), int(m.group(2)))]) if m else set()


def scan(path):
    out = []
    func = None
    lines = open(path).read().split("\n")
    n = len(lines)
    i = 0
    then_start = 0
    masked = False        # inside the `then` side of an if (exec narrowed by s_and_saveexec and not yet restored)
    while i < n:
        l = lines[i]
        m = FUNC.match(l)
        if m and not l.startswith("."):
            func = m.group(1)
        if EXEC_MASKED_ENTRY.match(l):
            masked = True
            then_start = i
        elif re.match(r"^\s*s_or_b64 exec, exec", l):
            masked = False
        elif EXEC_FLIP.match(l):
            # look back to the start of this block: any vector copy / spill between the block label and the flip?
            j = i - 1
            hits = []
            while j >= 0 and not LABEL.match(lines[j].strip() if lines[j].startswith(";") else lines[j]):
                if VEC_WRITE.match(lines[j]):
                    hits.append(lines[j].strip())
                j -= 1
            if hits and masked:
                # a PHI copy (value defined on the `then` side moved to the merge register) has its twin on the `else` side;
                # keep only copies whose destination the else side never writes
                k = i + 1
                target = None
                while k < min(n, i + 6):
                    mm = re.match(r"^\s*s_cbranch_execz\s+(\.LBB\d+_\d+)", lines[k])
                    if mm:
                        target = mm.group(1) + ":"
                        break
                    k += 1
                else_txt = []
                if target:
                    k += 1
                    while k < n and not lines[k].startswith(target):
                        else_txt.append(lines[k])
                        k += 1
                else_blob = "\n".join(else_txt)
                keep = []
                for h in hits:
                    mm = re.match(r"^(v_mov_b(?:32|64)_e32|v_accvgpr_write_b32|v_accvgpr_read_b32|v_pk_mov_b32)\s+([av]\[?\d+(?::\d+)?\]?),\s*([av]\[?\d+(?::\d+)?\]?)", h)
                    if not mm:
                        continue       # immediates / spills: not a register-to-register save
                    dst = mm.group(2)
                    regs = regset(dst)
                    written = False
                    for el in else_txt:
                        m2 = re.match(r"^\s*\S+\s+([av]\[?\d+(?::\d+)?\]?)", el)
                        if m2 and regset(m2.group(1)) & regs:
                            written = True
                            break
                    # ... and whose SOURCE the then side did not compute: a value that lives through the whole if-else, i.e. a
                    # live-range split copy, not the then side's contribution to a merge
                    src = regset(mm.group(3))
                    computed = False
                    for tl in lines[then_start:i]:
                        m2 = re.match(r"^\s*(\S+)\s+([av]\[?\d+(?::\d+)?\]?)", tl)
                        if m2 and regset(m2.group(2)) & src and tl.strip() != h:
                            computed = True
                            break
                    # ... and which is undone later (src <- dst): the signature of a save / restore pair around a region that
                    # re-uses src.  (A merge register initialised before the if and overwritten by the then side is not.)
                    restored = False
                    for rl in lines[i:min(n, i + 6000)]:
                        if rl.startswith(".Lfunc_end"):
                            break
                        m3 = re.match(r"^\s*(v_mov_b(?:32|64)_e32|v_accvgpr_write_b32|v_accvgpr_read_b32|v_pk_mov_b32)\s+([av]\[?\d+(?::\d+)?\]?),\s*([av]\[?\d+(?::\d+)?\]?)", rl)
                        if m3 and m3.group(2) == mm.group(3) and m3.group(3) == mm.group(2):
                            restored = True
                            break
                    if not written and not computed and restored and target:
                        keep.append(h)
                # spill stores before the flip: scratch_store_dword* off, v[a:b], <base> offset:N of a register the then side did
                # not compute, into a slot the else side does not store to, reloaded from that slot later
                for h in hits:
                    ms = re.match(r"^(?:scratch|buffer)_store_dword(?:x[234])?\s+(?:off,\s*)?(v\[?\d+(?::\d+)?\]?),\s*(.*)$", h)
                    if not ms:
                        continue
                    src = regset(ms.group(1))
                    slot = re.sub(r"\s+", " ", ms.group(2)).strip()
                    computed = False
                    for tl in lines[then_start:i]:
                        m2 = re.match(r"^\s*(\S+)\s+([av]\[?\d+(?::\d+)?\]?)", tl)
                        if m2 and not m2.group(1).startswith(("scratch_store", "buffer_store")) and regset(m2.group(2)) & src:
                            computed = True
                            break
                    stored_else = any(re.match(r"^\s*(?:scratch|buffer)_store_dword", el) and re.sub(r"\s+", " ", el).strip().endswith(slot) for el in else_txt)
                    reloaded = False
                    for rl in lines[i:min(n, i + 6000)]:
                        if rl.startswith(".Lfunc_end"):
                            break
                        m3 = re.match(r"^\s*(?:scratch|buffer)_load_dword(?:x[234])?\s+(v\[?\d+(?::\d+)?\]?),\s*(?:off,\s*)?(.*)$", rl)
                        if m3 and re.sub(r"\s+", " ", m3.group(2)).strip() == slot:
                            reloaded = True
                            break
                    if not computed and not stored_else and reloaded and target:
                        keep.append(h)
                if keep:
                    out.append((func, i + 1, keep))
            masked = True      # now the else side
        i += 1
    return out


if __name__ == "__main__":
    bad = 0
    for p in sys.argv[1:]:
        for func, line, hits in scan(p):
            bad += 1
            print("%s:%d %s: %d vector copies before the exec flip of a flow block: %s" % (p, line, (func or "?")[:70], len(hits), "; ".join(hits[:4])))
    print("suspicious flow-block copies:", bad)
    sys.exit(1 if bad else 0)
