#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of tools/profile_round.sh into the files kept under profiles/:
<tag>_bench_kernel_stats.csv, <tag>_field_pmc.json, <tag>_bench_profile.md."""
import collections, csv, glob, json, os, shutil, sys

out, tag = sys.argv[1], sys.argv[2]
PASSES = 9            # 2 allocator-priming passes + --warmup 2 + --steps 5


def one(pattern):
    f = glob.glob(os.path.join(out, pattern), recursive=True)
    return f[0] if f else None


def last_json(path):
    for line in reversed(open(path).read().splitlines()):
        if line.startswith("{"):
            return line
    return ""


trace = one("trace/**/*kernel_trace.csv")
acc = collections.OrderedDict()
for r in csv.DictReader(open(trace)):
    acc.setdefault(r["Kernel_Name"], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in acc.values())
stats = one("trace/**/*kernel_stats.csv")
if stats:
    shutil.copy(stats, os.path.join(out, "%s_bench_kernel_stats.csv" % tag))


def pmc(sub, name):
    f = one(sub + "/**/*counter_collection.csv")
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name:
            d[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in d.items()}


fetch, write = pmc("pmc_fetch", "FETCH_SIZE"), pmc("pmc_write", "WRITE_SIZE")


def find(d, sub):
    for k, v in d.items():
        if sub in k:
            return v
    return None


fk = "field_tile_kernel"
fkb, wkb = find(fetch, fk), find(write, fk)
field_us = [sum(v) / len(v) for k, v in acc.items() if fk in k][0]
import hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pmcj = {"kernel": fk, "command": "python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras",
        # bench.py uses these counter bytes only for the build of the kernel they were measured on
        "field_hip_sha256": hashlib.sha256(open(os.path.join(ROOT, "tomography_3d_reconstructor_amd", "csrc", "field.hip"), "rb").read()).hexdigest(),
        "fetch_size_kb_raw": fkb, "write_size_kb": wkb, "hbm_bytes_per_launch": (2 * fkb + wkb) * 1024,
        "correction": "reads x2 (gfx950 FETCH_SIZE counts half of the bytes of coalesced reads; calibrated in round 1 on "
                      "pack16_kernel's exact 1 GiB -- pack_close_kernel reads 34/32 of it), writes exact", "kernel_us_in_trace": field_us}
json.dump(pmcj, open(os.path.join(out, "%s_field_pmc.json" % tag), "w"), indent=1)

md = ["# Round profile `%s` -- `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras`\n" % tag,
      "MI355X (gfx950), 1024^3 ellipsoid, 9 passes of the hot path (2 priming + 2 warm-up + 5 timed).  Full CSV: `%s_bench_kernel_stats.csv`.\n" % tag,
      "Bench line of the profiled run: `%s`\n" % last_json(os.path.join(out, "bench_trace.log")),
      "Bench line, un-profiled (`python bench.py`): `%s`\n" % last_json(os.path.join(out, "bench_plain.log")),
      "\n| kernel | calls | avg us | ms per pass | % |\n|---|---|---|---|---|"]
for n, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    md.append("| `%s` | %d | %.2f | %.3f | %.2f |" % (n[:70], len(v), sum(v) / len(v), sum(v) / PASSES / 1e3, 100 * sum(v) / tot))
md.append("\n(`at::native::*` kernels are the synthetic-mask generation before the timed region.  The default command puts the front of a pass -- "
          "pack + close + smoothing -- on a second stream behind the field kernel of the pass before: kernels that then run next to each other "
          "are longer in this table than alone.)\n")
trace1 = one("trace1/**/*kernel_trace.csv")
if trace1:
    acc1 = collections.OrderedDict()
    for r in csv.DictReader(open(trace1)):
        acc1.setdefault(r["Kernel_Name"], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    tot1 = sum(sum(v) for k, v in acc1.items() if not k.startswith(("void at::", "void (anonymous", "__amd")))
    md.append("\n## The same passes with `--one-stream` (every kernel alone on the GPU): avg us per launch\n")
    md.append("Bench line: `%s`\n" % last_json(os.path.join(out, "bench_trace1.log")))
    md.append("| kernel | calls | avg us | ms per pass |\n|---|---|---|---|")
    for n, v in sorted(acc1.items(), key=lambda kv: -sum(kv[1])):
        if n.startswith(("void at::", "void (anonymous", "__amd")):
            continue
        md.append("| `%s` | %d | %.2f | %.3f |" % (n[:70], len(v), sum(v) / len(v), sum(v) / PASSES / 1e3))
    md.append("\nhot-path kernels per pass: %.3f ms\n" % (tot1 / PASSES / 1e3))
    s1 = one("trace1/**/*kernel_stats.csv")
    if s1:
        shutil.copy(s1, os.path.join(out, "%s_bench_kernel_stats_one_stream.csv" % tag))
md.append("\n## PMC passes (separate runs of the same command: `--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`), per dispatch, KB\n")
md.append("Calibration (round 1): `pack16_kernel` reads exactly 1 GiB (1 048 576 KB) with 16 B/lane loads; its raw FETCH_SIZE shows the 1/2 factor "
          "the MI355X guide documents for coalesced reads on gfx950, so reads are doubled below (`pack_close_ho_kernel`, round 4, reads 130/128 GiB: only the outer waves of a workgroup "
          "re-read a neighbour's slice; rounds 1-3: 34/32).  WRITE_SIZE is exact.\n")
md.append("| kernel | FETCH_SIZE KB (raw) | WRITE_SIZE KB | HBM bytes = 2*FETCH + WRITE |\n|---|---|---|---|")
for sub in ("pack_close_ho_kernel", "pack_close_kernel", "morph_wave32_kernel<4, 6", "morph_wave32_kernel<4, 5", fk, "mc_classify_bits_kernel", "mc3_list_kernel", "mc3_eval_kernel",
            "mc3_vertices_kernel", "uq3_sortrank_kernel", "rocprim", "uq3_rank_kernel", "mc3_faces_kernel"):
    a, b = find(fetch, sub), find(write, sub)
    if a is not None and b is not None:
        md.append("| `%s` | %.0f | %.0f | %.3e |" % (sub, a, b, (2 * a + b) * 1024))
alg = 5.0 * 1026 ** 3
md.append("\nField (\"SDF\") kernel `%s`: avg %.1f us per launch in this trace -> algorithmic 5 B x 1026^3 = %.3e B / launch = %.0f GB/s "
          "= %.1f %% of the 8 TB/s HBM3E peak (SURVEY 8(d)'s accounting: 1 B mask + 4 B field per padded voxel); measured HBM traffic %.3e B per launch "
          "= %.0f GB/s = %.1f %% of peak by the bytes the kernel moves (it reads the bit-packed volume; the 1 B/voxel mask is read by pack_close_kernel).\n"
          % (fk, field_us, alg, alg / field_us / 1e3, alg / field_us / 1e3 / 80.0, pmcj["hbm_bytes_per_launch"],
             pmcj["hbm_bytes_per_launch"] / field_us / 1e3, pmcj["hbm_bytes_per_launch"] / field_us / 1e3 / 80.0))
open(os.path.join(out, "%s_bench_profile.md" % tag), "w").write("\n".join(md))
print("\n".join(md[-3:]))
