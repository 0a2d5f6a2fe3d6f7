"""Developer tool: static instruction mix per kernel of a translation unit (gfx950 assembly from hipcc -S).
   python tools/isa_stats.py touch_gs_amd/csrc/raster.hip [extra hipcc flags]"""
import collections, re, subprocess, sys, os
src = sys.argv[1]
here = os.path.dirname(os.path.abspath(src))
flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", "-S", "--cuda-device-only", "-I", here] + sys.argv[2:]
asm = subprocess.run(["/opt/rocm/bin/hipcc", *flags, src, "-o", "-"], capture_output=True, text=True).stdout
cur, body = None, collections.defaultdict(list)
for l in asm.split("\n"):
    m = re.match(r"^(_Z\w+):", l)
    if m:
        cur = m.group(1); continue
    if cur and l.startswith("\t") and not l.strip().startswith((".", ";")):
        op = l.strip().split()[0]
        body[cur].append(op)
        if op == "s_endpgm":
            cur = None
meta = dict(re.findall(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)", asm))
for name, ins in body.items():
    c = collections.Counter(ins)
    f = lambda p: sum(v for k, v in c.items() if k.startswith(p))
    short = re.sub(r"^_ZN?\d*(_GLOBAL__N_1)?\d*", "", name)[:48]
    print(f"{short:48s} total {len(ins):5d} valu {f('v_'):5d} salu {f('s_'):5d} ds {f('ds_'):4d} | exp {f('v_exp')} rcp {f('v_rcp')} "
          f"cndmask {f('v_cndmask')} cmp {f('v_cmp')} fma {f('v_fma') + f('v_fmac')} branch {f('s_cbranch')}")
for m in re.finditer(r"\.name:\s+(\S+)(.*?)\.wavefront_size", asm, re.S):
    v = re.search(r"\.vgpr_count:\s+(\d+)", m.group(2)); a = re.search(r"\.agpr_count:\s+(\d+)", m.group(2))
    l = re.search(r"\.group_segment_fixed_size:\s+(\d+)", m.group(2)); sp = re.search(r"\.vgpr_spill_count:\s+(\d+)", m.group(2))
    print(f"  {m.group(1)[:70]:70s} vgpr {v.group(1) if v else '?'} agpr {a.group(1) if a else '?'} lds {l.group(1) if l else '?'} spill {sp.group(1) if sp else '?'}")
