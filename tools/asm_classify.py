"""tools/asm_classify.py ASM KERNEL_SUBSTRING [--by line|func] [--top N]: static instruction mix of one kernel of a gfx950 assembly file
built with -gline-tables-only (GRT_VARIANT_ASM=1 tools/build_variant.sh NAME "-DGRT_FAST_BUILD -gline-tables-only"): instructions by
class (fp64 FMA / MUL / ADD, fp64 other, fp32 arithmetic, conversions, moves / selects, compares, integer / address, transcendental
seeds, LDS, VMEM, SALU, ...) and by the source function each instruction's .loc line falls in (innermost inlined location).  Static
counts: a loop body counts once (the transfer kernel's step body has no inner loops at FB = 1 apart from table searches)."""
import re, sys, collections, os, bisect
asm, sub = sys.argv[1], sys.argv[2]
by = "func"; top = 40
if "--by" in sys.argv: by = sys.argv[sys.argv.index("--by") + 1]
if "--top" in sys.argv: top = int(sys.argv[sys.argv.index("--top") + 1])

def klass(op):
    if op.startswith("v_fma_f64") or op.startswith("v_fmac_f64"): return "f64 fma"
    if op.startswith("v_mul_f64"): return "f64 mul"
    if op.startswith("v_add_f64"): return "f64 add"
    if op.startswith(("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64")): return "f64 seed"
    if re.match(r"v_(div_scale|div_fmas|div_fixup|ldexp|frexp|trig_preop|rndne|floor|fract|ceil|trunc|max|min|cmp\w*)_f64", op) or op.startswith("v_cmp") and "f64" in op: return "f64 other"
    if "_f64" in op and op.startswith("v_cvt"): return "cvt"
    if "_f64" in op: return "f64 other"
    if op.startswith("v_cvt"): return "cvt"
    if op.startswith("v_pk_"): return "f32 packed"
    if re.match(r"v_(fma|fmac|mul|add|sub|subrev|mad|mac)_f32", op): return "f32 arith"
    if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_f32", op): return "f32 seed"
    if "_f32" in op: return "f32 other"
    if op.startswith(("v_mov", "v_cndmask", "v_accvgpr", "v_swap")): return "mov/select"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane", "v_permlane", "v_bpermute")): return "lane ops"
    if op.startswith("v_cmp"): return "int cmp"
    if op.startswith("v_"): return "int/addr/logic"
    if op.startswith("ds_"): return "LDS"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")): return "VMEM"
    if op.startswith("s_load") or op.startswith("s_buffer_load"): return "SMEM"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"): return "wait/nop"
    if op.startswith("s_"): return "SALU"
    return "other"

files = {}
# map (file, line) -> function name by scanning the sources for function headers (GRT_HD / GRT_NI / __global__ / template lines)
def func_index(path):
    heads = []
    try:
        src = open(path).read().splitlines()
    except Exception:
        return [], []
    pat = re.compile(r"^\s*(?:template\s*<[^>]*>\s*)?(?:GRT_ALWAYS_INLINE\s+)?(?:GRT_HD|GRT_NI|__global__|__device__|static)\b.*?\b([A-Za-z_]\w*)\s*\(")
    for i, l in enumerate(src, 1):
        m = pat.match(l)
        if m and not l.strip().startswith("//"):
            heads.append((i, m.group(1)))
    return [h[0] for h in heads], [h[1] for h in heads]

cur = None; inside = False
cls = collections.Counter(); where = collections.Counter(); wherecls = collections.defaultdict(collections.Counter)
fidx = {}
loc = (None, 0)
for line in open(asm, errors="replace"):
    s = line.strip()
    m = re.match(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', s)
    if m:
        files[int(m.group(1))] = os.path.join(m.group(2), m.group(3)) if m.group(3) else m.group(2)
        continue
    if s.startswith(".type") and "@function" in s:
        name = s.split()[1].rstrip(",")
        inside = sub in name
        if inside: cur = name
        continue
    if not inside: continue
    if s.startswith(".Lfunc_end"): inside = False; continue
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
    if m:
        # library headers (fma, sqrt, log ... wrappers of __clang_hip_math.h, the shuffle / atomic headers) are transparent: their
        # instructions go to the last location in the project's own sources
        fn = files.get(int(m.group(1)), "")
        if "grtrans_amd" in fn or "csrc" in fn or not fn.startswith("/"): loc = (int(m.group(1)), int(m.group(2)))
        continue
    if not s or s.startswith((".", ";", "//")) or s.endswith(":"): continue
    op = s.split()[0]
    if not re.match(r"^[a-z_0-9]+$", op): continue
    k = klass(op)
    cls[k] += 1
    f = files.get(loc[0], "?")
    if by == "line": key = "%s:%d" % (os.path.basename(f), loc[1])
    else:
        if f not in fidx: fidx[f] = func_index(f)
        ln, nm = fidx[f]
        j = bisect.bisect_right(ln, loc[1]) - 1
        key = "%s:%s" % (os.path.basename(f), nm[j] if j >= 0 else "?")
    where[key] += 1; wherecls[key][k] += 1
print("kernel:", cur)
tot = sum(cls.values())
valu = sum(v for k, v in cls.items() if k not in ("LDS", "VMEM", "SMEM", "wait/nop", "SALU", "other"))
print("static instructions %d, VALU %d" % (tot, valu))
for k, v in cls.most_common(): print("  %-16s %6d  %5.1f %%" % (k, v, 100.0 * v / tot))
# r06: estimated SIMD cycles by the measured cost model (tools/experiments/dp_rate_bench.hip, sgpr_const_bench.hip): an fp64 / fp32 /
# integer vector instruction 4 cycles, a transcendental seed (v_rcp / v_rsq / v_sqrt / v_exp / v_log, fp64 or fp32) 16, a register move ~2;
# scalar, memory and wait instructions issue beside the vector pipe
W = collections.defaultdict(lambda: 4, {"f64 seed": 16, "f32 seed": 16, "mov/select": 3, "LDS": 0, "VMEM": 0, "SMEM": 0, "wait/nop": 0, "SALU": 0, "other": 0})
cyc = {key: sum(W[k] * n for k, n in wherecls[key].items()) for key in where}
print("estimated vector-pipe cycles (static): %d" % sum(cyc.values()))
print("by source %s (top %d):" % (by, top))
order = ["f64 fma", "f64 mul", "f64 add", "f64 seed", "f64 other", "f32 arith", "f32 packed", "f32 seed", "f32 other", "cvt", "mov/select", "int cmp", "int/addr/logic", "lane ops", "LDS", "VMEM", "SMEM", "SALU", "wait/nop"]
print("  %-44s %6s %6s | %s" % ("", "cycles", "all", " ".join("%5s" % o.replace("f64 ", "d.").replace("f32 ", "s.").replace("mov/select", "mov").replace("int/addr/logic", "int").replace("wait/nop", "wait")[:5] for o in order)))
for key, c in sorted(cyc.items(), key=lambda kv: -kv[1])[:top]:
    print("  %-44s %6d %6d | %s" % (key[:44], c, where[key], " ".join("%5d" % wherecls[key][o] for o in order)))
