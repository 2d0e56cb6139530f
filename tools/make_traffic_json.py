"""profiles/xcorr_traffic.json from two PMC summaries (tools/rocpd_pmc.py --md of measure/gpu_pmc.sh at 30 and 100 tracks):
HBM bytes per launch of the graded kernel = 2 x FETCH_SIZE + WRITE_SIZE (KiB per dispatch; the factor 2 is the gfx950 correction
of MI355X_MICROARCH.md, calibrated on this kernel in profiles/r02_fetch_calibration.md), stamped with the kernel's name and the hash
of the sources it was compiled from — bench.py refuses the numbers for any other kernel or source state.

    python tools/make_traffic_json.py profiles/r04_pmc_counters_n30.md profiles/r04_pmc_counters_n100.md
"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import siammot_amd.ops as ops


def kernel_counters(md, kernel_prefix):
    sec, out = None, {}
    for line in open(md):
        if line.startswith("### "):
            sec = line
        elif sec and kernel_prefix in sec.replace(" ", "") and line.startswith("- "):
            m = re.match(r"- (\w+): ([-+0-9.e]+)", line)
            if m:
                out[m.group(1)] = float(m.group(2))
    return out


def main():
    md30, md100 = sys.argv[1], sys.argv[2]
    name = ops.fused_kernel_name()                       # sr_xcorr_fused9_kernel<30,15,2,true>
    prefix = name.replace(" ", "").rstrip(">")           # rocprofv3 prints further template arguments behind these
    res = {}
    for n, md in (("30", md30), ("100", md100)):
        c = kernel_counters(md, prefix)
        if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
            raise SystemExit("%s: no FETCH_SIZE / WRITE_SIZE for %s" % (md, name))
        res[n] = int(round((2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0))
        if "SQ_INSTS_VALU" in c:             # wave-level vector instructions per launch: bench.py's roofline.valu_issue
            res.setdefault("valu_insts", {})[n] = int(round(c["SQ_INSTS_VALU"]))
    res["kernel"] = name
    res["source_sha1"] = bench.fused_source_sha1()
    res["source"] = ("%s / %s: 2 x FETCH_SIZE + WRITE_SIZE (KiB) of %s, separate --pmc passes over bench.py --tracks N (8 rotating "
                     "feature sets, 307 MB; measure/gpu_pmc.sh); the factor 2 on FETCH_SIZE is the gfx950 correction of "
                     "MI355X_MICROARCH.md, calibrated on this kernel's access pattern in profiles/r02_fetch_calibration.md (requests are "
                     "128-byte lines tallied at 64 B; WRITE_SIZE is exact)" % (os.path.relpath(md30, ROOT), os.path.relpath(md100, ROOT), name))
    json.dump(res, open(os.path.join(ROOT, "profiles", "xcorr_traffic.json"), "w"))
    print(json.dumps(res)[:300])


if __name__ == "__main__":
    main()
