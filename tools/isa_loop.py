#!/usr/bin/env python3
"""Instruction mix of one tCG step of a kernel in libgraphik_amd.so: the code between two consecutive reductions
(v_permlane32_swap groups) of the role-swapped loop body.

    tools/isa_loop.py <mangled-substring> [-v] [--min-group N]

The library holds one code object per translation unit (gik_k_*.hip); every one of them is searched."""
import collections, os, re, subprocess, sys, tempfile
lib = os.environ.get("GIK_LIB_PATH") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "graphik_amd", "lib",
                                                     "libgraphik_amd.so")
args = [a for a in sys.argv[1:] if not a.startswith("-")]
pat = args[0] if args else "rtr_npt_kernelILi1"
min_group = int(sys.argv[sys.argv.index("--min-group") + 1]) if "--min-group" in sys.argv else 8
tmp = tempfile.mkdtemp()
subprocess.check_call(["cp", lib, tmp + "/lib.so"])
subprocess.check_call(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", "lib.so"], cwd=tmp, stdout=subprocess.DEVNULL)
body = None
for co in sorted(f for f in os.listdir(tmp) if "gfx950" in f):
    txt = subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", co], cwd=tmp).decode().split("\n")
    starts = [i for i, l in enumerate(txt) if re.match(r"^[0-9a-f]+ <.*" + pat, l)]
    if starts:
        start = starts[0]
        end = next((i for i in range(start + 1, len(txt)) if re.match(r"^[0-9a-f]+ <", txt[i])), len(txt))
        body = txt[start:end]
        print(co, txt[start].split("<")[1].rstrip(">:"))
        break
if body is None:
    sys.exit("no kernel matches " + pat)
idx = [i for i, l in enumerate(body) if "v_permlane32_swap" in l]
groups, cur = [], [idx[0]]
for a in idx[1:]:
    if a - cur[-1] < 80: cur.append(a)
    else: groups.append(cur); cur = [a]
groups.append(cur)
big = [g for g in groups if len(g) >= min_group]
print("swap groups (first line, last line, swaps):", [(g[0], g[-1], len(g)) for g in groups])
seg = body[big[0][0]:big[1][0]]
c = collections.Counter()
for l in seg:
    p = l.split()
    if p and re.match(r"^[a-z_0-9]+$", p[0]): c[p[0]] += 1
print("static instructions in one step:", sum(c.values()))
cls = collections.Counter()
for k, v in c.items():
    key = ("fp64" if re.search(r"_f64", k) else "accvgpr" if "accvgpr" in k else "readlane" if "readlane" in k else
           "permlane" if "permlane" in k else "dpp" if "dpp" in k else "ds" if k.startswith("ds_") else
           "salu" if k.startswith("s_") else "mov" if k.startswith("v_mov") else "cndmask" if "cndmask" in k else "valu-other")
    cls[key] += v
print(dict(cls))
if "-v" in sys.argv:
    for k, v in c.most_common(50): print(f"  {k:30s} {v}")
if "--dump" in sys.argv:
    print("\n".join(seg))
