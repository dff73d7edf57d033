"""Condense the rocprofv3 CSV output of tools/collect_profiles.sh (gpurun_out/r03) into the tracked files under
profiles/: per-kernel duration statistics, per-kernel counter sums / per-launch averages, the MFMA-busy fractions and
the HBM-side traffic per launch that bench.py reads (profiles/r03_pmc_traffic.json)."""
import csv
import glob
import json
import os
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = os.environ.get("MBFIR_ROUND", "r06")
SRC = os.path.join(ROOT, "gpurun_out", ROUND)
DST = os.environ.get("MBFIR_PROFILE_DST", os.path.join(ROOT, "profiles"))


def find_db(d):
    hits = glob.glob(os.path.join(SRC, d, "**", "*_results.db"), recursive=True)
    return hits[0] if hits else None


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("mbfir::", "")
    return name.split("(")[0].replace("void ", "")


def kernel_stats(d, out):
    """rocprofv3's default output is an SQLite database; `kernels` is its view of the kernel dispatches."""
    path = find_db(d)
    if not path:
        return None
    import sqlite3
    cur = sqlite3.connect(path).cursor()
    agg = defaultdict(lambda: [0, 0.0])
    for name, start, end in cur.execute("select name, start, end from kernels"):
        k = short(name)
        agg[k][0] += 1
        agg[k][1] += (end - start) / 1e3
    tot = sum(v[1] for v in agg.values())
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    with open(os.path.join(DST, out), "w") as fh:
        fh.write("kernel,calls,total_us,avg_us,percent\n")
        for k, (c, t) in rows:
            fh.write('"%s",%d,%.1f,%.3f,%.2f\n' % (k, c, t, t / c, 100 * t / tot))
    return {k: (c, t) for k, (c, t) in rows}


def counter_sums(d):
    path = find_db(d)
    if not path:
        return {}
    import sqlite3
    cur = sqlite3.connect(path).cursor()
    agg = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(set)
    for name, cname, val, disp in cur.execute("select name, counter_name, counter_value, dispatch_id from pmc_events"):
        k = short(name)
        agg[k][cname] += float(val)
        calls[k].add(disp)
    return {k: dict(v, calls=len(calls[k])) for k, v in agg.items()}


def main():
    os.makedirs(DST, exist_ok=True)
    commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    for d, out in (("bench_trace", ROUND + "_bench_kernel_stats.csv"), ("unit_trace", ROUND + "_unit16_kernel_stats.csv"),
                   ("unit8_trace", ROUND + "_unit8_kernel_stats.csv"), ("dense_trace", ROUND + "_dense_kernel_stats.csv"),
                   ("hetero_trace", ROUND + "_hetero64_kernel_stats.csv"), ("c3_trace", ROUND + "_c3_batch8_kernel_stats.csv"),
                   ("c3u_trace", ROUND + "_c3_units16_kernel_stats.csv")):
        kernel_stats(d, out)
    if os.path.exists(os.path.join(SRC, "bench.json")):
        with open(os.path.join(SRC, "bench.json")) as fh, open(os.path.join(DST, ROUND + "_bench.json"), "w") as out:
            out.write(fh.read())
    report = {"commit": commit}
    for tag, lanes in (("unit", 16), ("dense", 1)):
        busy, insts = counter_sums(tag + "_pmc_busy"), counter_sums(tag + "_pmc_insts")
        fetch, write = counter_sums(tag + "_pmc_fetch"), counter_sums(tag + "_pmc_write")
        with open(os.path.join(DST, ROUND + "_pmc_mfma_%s.csv" % tag), "w") as fh:
            fh.write("kernel,calls,SQ_VALU_MFMA_BUSY_CYCLES,SQ_BUSY_CYCLES,SQ_WAVE_CYCLES,GRBM_GUI_ACTIVE,mfma_busy_over_sq_busy,"
                     "SQ_INSTS_VALU_MFMA_MOPS_F64,SQ_INSTS_VALU_MFMA_F64,SQ_INSTS_VALU,FETCH_SIZE_per_launch,WRITE_SIZE_per_launch\n")
            for k in sorted(busy, key=lambda kk: -busy[kk].get("SQ_BUSY_CYCLES", 0)):
                b, i_, f, w = busy[k], insts.get(k, {}), fetch.get(k, {}), write.get(k, {})
                frac = b.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / b["SQ_BUSY_CYCLES"] if b.get("SQ_BUSY_CYCLES") else 0.0
                fh.write('"%s",%d,%.0f,%.0f,%.0f,%.0f,%.4f,%.0f,%.0f,%.0f,%.1f,%.1f\n' % (
                    k, b["calls"], b.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), b.get("SQ_BUSY_CYCLES", 0), b.get("SQ_WAVE_CYCLES", 0),
                    b.get("GRBM_GUI_ACTIVE", 0), frac, i_.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0), i_.get("SQ_INSTS_VALU_MFMA_F64", 0),
                    i_.get("SQ_INSTS_VALU", 0), f.get("FETCH_SIZE", 0) / max(1, f.get("calls", 1)), w.get("WRITE_SIZE", 0) / max(1, w.get("calls", 1))))
    finalize(DST, commit)


def finalize(dst, commit):
    """profiles/r03_pmc_traffic.json (read by bench.py) from the per-kernel CSVs: k_chol_step from the lock-step unit of
    8 designs, k_gram from the dense single design.  FETCH_SIZE / WRITE_SIZE are reported in KB; FETCH_SIZE is doubled
    for 16-B/lane streaming reads on gfx950 (MI355X_MICROARCH.md, HBM section).  MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES
    (summed over the 1024 SIMDs) / (GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 * 1024)."""
    report = {"commit": commit}
    for tag, kern, key, lanes in (("unit", "k_chol_dag", "k_chol", 16), ("dense", "k_gram", "k_gram", 1)):
        path = os.path.join(dst, ROUND + "_pmc_mfma_%s.csv" % tag)
        if not os.path.exists(path):
            continue
        with open(path) as fh:
            for r in csv.DictReader(fh):
                if r["kernel"] == kern or r["kernel"].startswith(kern + "<"):
                    calls = float(r["calls"])
                    fb = float(r["FETCH_SIZE_per_launch"]) * 1024 * 2
                    wb = float(r["WRITE_SIZE_per_launch"]) * 1024
                    gui = float(r["GRBM_GUI_ACTIVE"])
                    report["%s_bytes_per_launch" % key] = fb + wb
                    report["%s_fetch_x2_bytes" % key] = fb
                    report["%s_write_bytes" % key] = wb
                    report["%s_mfma_util" % key] = float(r["SQ_VALU_MFMA_BUSY_CYCLES"]) / (gui / 8 * 1024) if gui else None
                    report["%s_mfma_flop_per_launch" % key] = float(r["SQ_INSTS_VALU_MFMA_MOPS_F64"]) * 512 / calls
                    report["%s_launches_profiled" % key] = int(calls)
                    if key == "k_chol":
                        report["lanes"] = lanes
                    break
    with open(os.path.join(dst, ROUND + "_pmc_traffic.json"), "w") as fh:
        json.dump(report, fh, indent=1)
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "finalize":
        # local step: copy the box's summaries (gpurun_out/r03_profiles) into profiles/ and stamp the commit
        import shutil
        src = os.path.join(ROOT, "gpurun_out", ROUND + "_profiles")
        dst = os.path.join(ROOT, "profiles")
        for f in glob.glob(os.path.join(src, ROUND + "_*")):
            shutil.copy(f, dst)
        commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
        finalize(dst, commit)
    else:
        main()
