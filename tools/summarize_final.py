#!/usr/bin/env python3
"""Turn gpurun_out/final/ (tools/prof_final.sh) into the committed summaries under profiles/."""
import csv, collections, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "gpurun_out", "final")
tag = sys.argv[1] if len(sys.argv) > 1 else "r02_final"
out = []
out.append("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras   (7 proofs of miden:20:51:8: 1 warm-up, 3 timed, 3 of the breakdown pass; ns)")
out.append("# (k_perm_rate = the register-only Poseidon2 rate bench.py reports next to roofline_valu; it runs once, outside the timed region)")
out.append(open(os.path.join(O, "kt", "kt_kernel_stats.csv")).read().strip())


def pmc(name):
    rows = list(csv.DictReader(open(os.path.join(O, f"pmc_{name}", "pmc_counter_collection.csv"))))
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in rows:
        k = r["Kernel_Name"].split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    return agg, {k: len(v) for k, v in disp.items()}


per_launch = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg, disp = pmc(c)
    out.append(f"# rocprofv3 --pmc {c} (its own pass, bench.py --steps 3 --warmup 0: 6 proofs), unit KB as reported\nkernel,dispatches,sum_KB,per_dispatch_KB")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][c])[:10]:
        out.append(f"{k},{disp[k]},{v[c]:.0f},{v[c] / disp[k]:.1f}")
        per_launch.setdefault(k, {})[c] = v[c] / disp[k]
agg, disp = pmc("SQ")
out.append("# rocprofv3 --pmc SQ_* (own pass, 6 proofs)\nkernel,dispatches,waves,valu_insts,valu_per_wave,SQ_ACTIVE_INST_VALU/SQ_BUSY_CYCLES(raw ratio: compare kernels with each other),wait_inst_any/wave_cycles,wait_any/wave_cycles")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["SQ_INSTS_VALU"])[:8]:
    w = max(1.0, v["SQ_WAVES"])
    out.append(f"{k},{disp[k]},{int(w)},{v['SQ_INSTS_VALU']:.3e},{v['SQ_INSTS_VALU'] / w:.0f},{v['SQ_ACTIVE_INST_VALU'] / max(1, v['SQ_BUSY_CYCLES']):.3f},"
               f"{v['SQ_WAIT_INST_ANY'] / max(1, v['SQ_WAVE_CYCLES']):.3f},{v['SQ_WAIT_ANY'] / max(1, v['SQ_WAVE_CYCLES']):.3f}")
la_sq = agg.get("k_leaf_absorb")
valu_per_perm = None
if la_sq:
    n_proofs = disp["k_leaf_absorb"] // 3  # three leaf launches per proof (main, aux, quotient)
    valu_per_perm = la_sq['SQ_INSTS_VALU'] / (n_proofs * (8 << 20) * 11 / 64)
    perms = n_proofs * (8 << 20) * 11  # the proofs of the pass, 8 * 2^20 leaves, 7 + 2 + 2 permutations per leaf
    out.append(f"# k_leaf_absorb: SQ_INSTS_VALU per permutation = {la_sq['SQ_INSTS_VALU'] * 64 / perms:.0f} wave-instructions x 64 lanes / {perms} permutations "
               f"= {la_sq['SQ_INSTS_VALU'] / (perms / 64):.0f} VALU instructions per permutation (loads/stores and address arithmetic of the kernel included)")
bench = open(os.path.join(O, "bench.json")).read().strip().splitlines()[-1]
out.append("# bench line (same build, un-profiled, python bench.py --steps 20 --warmup 5):")
out.append(bench)
open(os.path.join(ROOT, "profiles", f"{tag}_rocprof.txt"), "w").write("\n".join(out) + "\n")
la = per_launch.get("k_leaf_absorb", {})
if la:
    j = {"kernel": "lmcs_leaf_absorb", "device_kernel": "k_leaf_absorb", "fetch_size_kb_per_launch": la["FETCH_SIZE"], "write_size_kb_per_launch": la["WRITE_SIZE"],
         "fetch_correction": 2.0, "hbm_bytes_per_launch": (2.0 * la["FETCH_SIZE"] + la["WRITE_SIZE"]) * 1024,
         "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, averaged over the 3 leaf-absorb launches of a proof "
                 "(main, aux, quotient); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts wide coalesced reads at half)",
         "valu_per_permutation_SQ_INSTS_VALU": valu_per_perm,
         "source": f"profiles/{tag}_rocprof.txt"}
    json.dump(j, open(os.path.join(ROOT, "profiles", tag.split("_")[0] + "_pmc_leaf_absorb.json"), "w"), indent=1)
print("\n".join(out[-3:])[:1500])
