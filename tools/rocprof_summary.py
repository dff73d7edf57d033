"""Condense the rocprofv3 CSV output of tools/collect_profiles.sh (gpurun_out/r02) into the tracked files under
profiles/: per-kernel duration statistics, per-kernel counter sums / per-launch averages, the MFMA-busy fractions and
the HBM-side traffic per launch that bench.py reads (profiles/r02_pmc_traffic.json)."""
import csv
import glob
import json
import os
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r02")
DST = os.environ.get("MBFIR_PROFILE_DST", os.path.join(ROOT, "profiles"))


def find_db(d):
    hits = glob.glob(os.path.join(SRC, d, "**", "*_results.db"), recursive=True)
    return hits[0] if hits else None


def short(name):
    name = name.replace("mbfir::", "")
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
    for d, out in (("bench_trace", "r02_bench_kernel_stats.csv"), ("unit_trace", "r02_unit8_kernel_stats.csv"),
                   ("dense_trace", "r02_dense_kernel_stats.csv")):
        kernel_stats(d, out)
    if os.path.exists(os.path.join(SRC, "bench.json")):
        with open(os.path.join(SRC, "bench.json")) as fh, open(os.path.join(DST, "r02_bench.json"), "w") as out:
            out.write(fh.read())
    report = {"commit": commit}
    for tag, lanes in (("unit", 8), ("dense", 1)):
        busy, insts = counter_sums(tag + "_pmc_busy"), counter_sums(tag + "_pmc_insts")
        fetch, write = counter_sums(tag + "_pmc_fetch"), counter_sums(tag + "_pmc_write")
        with open(os.path.join(DST, "r02_pmc_mfma_%s.csv" % tag), "w") as fh:
            fh.write("kernel,calls,SQ_VALU_MFMA_BUSY_CYCLES,SQ_BUSY_CYCLES,SQ_WAVE_CYCLES,GRBM_GUI_ACTIVE,mfma_busy_over_sq_busy,"
                     "SQ_INSTS_VALU_MFMA_MOPS_F64,SQ_INSTS_VALU_MFMA_F64,SQ_INSTS_VALU,FETCH_SIZE_per_launch,WRITE_SIZE_per_launch\n")
            for k in sorted(busy, key=lambda kk: -busy[kk].get("SQ_BUSY_CYCLES", 0)):
                b, i_, f, w = busy[k], insts.get(k, {}), fetch.get(k, {}), write.get(k, {})
                frac = b.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / b["SQ_BUSY_CYCLES"] if b.get("SQ_BUSY_CYCLES") else 0.0
                fh.write('"%s",%d,%.0f,%.0f,%.0f,%.0f,%.4f,%.0f,%.0f,%.0f,%.1f,%.1f\n' % (
                    k, b["calls"], b.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), b.get("SQ_BUSY_CYCLES", 0), b.get("SQ_WAVE_CYCLES", 0),
                    b.get("GRBM_GUI_ACTIVE", 0), frac, i_.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0), i_.get("SQ_INSTS_VALU_MFMA_F64", 0),
                    i_.get("SQ_INSTS_VALU", 0), f.get("FETCH_SIZE", 0) / max(1, f.get("calls", 1)), w.get("WRITE_SIZE", 0) / max(1, w.get("calls", 1))))
        for kern in ("k_chol_step", "k_gram<1>", "k_gram"):
            for k in busy:
                if k == kern or k.startswith(kern):
                    f, w = fetch.get(k, {}), write.get(k, {})
                    # FETCH_SIZE / WRITE_SIZE are reported in KB; FETCH_SIZE is doubled for 16-B/lane streaming reads on
                    # gfx950 (MI355X_MICROARCH.md, HBM section)
                    fb = f.get("FETCH_SIZE", 0) / max(1, f.get("calls", 1)) * 1024 * 2
                    wb = w.get("WRITE_SIZE", 0) / max(1, w.get("calls", 1)) * 1024
                    key = "k_chol_step" if kern == "k_chol_step" else "k_gram"
                    report["%s_bytes_per_launch" % key] = fb + wb
                    report["%s_fetch_x2_bytes" % key] = fb
                    report["%s_write_bytes" % key] = wb
                    report["%s_mfma_busy_over_sq_busy" % key] = (busy[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / busy[k]["SQ_BUSY_CYCLES"]
                                                                 if busy[k].get("SQ_BUSY_CYCLES") else None)
                    if key == "k_chol_step":
                        report["lanes"] = lanes
                    break
    with open(os.path.join(DST, "r02_pmc_traffic.json"), "w") as fh:
        json.dump(report, fh, indent=1)
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
