"""Histogram of instruction mnemonics per basic block of one kernel in a hipcc -S listing (static view of the hot loops).
usage: python tools/isa_hist.py file.s <mangled-substring> [min_block_len]"""
import sys, re, collections
path, key = sys.argv[1], sys.argv[2]
minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 40
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().split(":")[0].endswith(key.split()[-1]) or (l.startswith("_Z") and key in l))
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith("\t.end_amdhsa_kernel") or lines[i].startswith(".Lfunc_end"))
blocks, cur, name = [], [], "entry"
for l in lines[start + 1:end]:
    if re.match(r"^\.LBB\d+_\d+:", l):
        blocks.append((name, cur)); cur, name = [], l.split(":")[0]
    else:
        m = re.match(r"^\t([a-z_0-9]+)", l)
        if m and not m.group(1).startswith("."): cur.append(m.group(1))
blocks.append((name, cur))
def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith("s_waitcnt") or op.startswith("s_barrier") or op.startswith("s_nop"): return op
    if op.startswith("s_"): return "salu"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")): return "trans"
    if op.startswith("v_accvgpr"): return "accmov"
    return "valu"
for name, b in blocks:
    if len(b) < minlen: continue
    c = collections.Counter(cls(o) for o in b)
    print(f"== {name}: {len(b)} instrs  " + "  ".join(f"{k}={v}" for k, v in sorted(c.items(), key=lambda kv: -kv[1])))
    v = collections.Counter(o for o in b if cls(o) in ("valu", "trans", "accmov"))
    print("   " + "  ".join(f"{k}:{n}" for k, n in v.most_common(24)))
