#!/usr/bin/env python3
"""Counter passes over ONE bench step at HEAD, per dispatch.  Run on the GPU box (gpurun): `python3 tools/pmc_round.py [batch] [grid]`
(output files are tagged with the round, TAG below).

This process never touches the GPU: it starts `rocprofv3 --kernel-trace --pmc <pass> -- python3 bench.py ...` once per counter
pass (separate runs, --kernel-trace only: MI355X_MICROARCH.md "HBM" / "rocprofv3 PMC slots"), reads the counter_collection CSVs,
cuts the dispatch stream into steps at the gather / voxelise launch that opens a forward, labels every dispatch position of the
last step with the launch key bench.py's own profiling pass recorded (--dump-launch-order) and writes

    gpurun_out/<TAG>_pmc.json        - what bench.py prints as roofline.traffic / roofline.hbm (copy to profiles/<TAG>_pmc.json)
    gpurun_out/<TAG>_pmc_table.txt   - every library dispatch of one step with all counters (copy to profiles/)

Both carry the ABI version and the hash of sceneego_amd/csrc they were taken on (sceneego_amd/_lib.py: source_fingerprint);
bench.py refuses a record whose fingerprint differs from the checkout it runs in.
Unit corrections (guide): FETCH_SIZE is in KiB and reports half of the bytes of wide streaming reads on gfx950 (x2, re-checked on every run by
tools/diag/copy_calib, which is built here when missing); WRITE_SIZE in KiB as is.
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
TAG = "r06"
PASSES = [
    ["FETCH_SIZE"],
    ["WRITE_SIZE"],
    ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_WAVE_CYCLES"],
    ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"],
    ["GRBM_GUI_ACTIVE"],
    ["SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS"],      # round 4: how busy the vector ALUs are beside the MFMAs
    ["SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_ACTIVE_INST_VMEM", "SQ_INST_CYCLES_SALU"],
]
BENCH = ["python3", "bench.py", "--steps", "2", "--warmup", "1", "--streams", "1", "--no-cpu-baseline", "--no-parity",
         "--no-kernel-events", "--no-extras"]
OURS = ("conv3d_", "deconv3d_", "maxpool2", "pointwise_chain3", "softargmax_", "::gather_", "voxelize_", "splitk_reduce", "fft7_")
FRONT = ("conv3d_k7", "fft7_fwd")       # the 7^3 front layer: Winograd kernel, or pass 1 of the frequency-domain form


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", n)[:48]


def run_pass(counters, batch, G):
    tag = counters[0]
    d = os.path.join(OUT, TAG + "pmc_" + tag)
    shutil.rmtree(d, ignore_errors=True)
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", *counters, "--output-format", "csv", "-d", d, "--"] + BENCH + \
          ["--batch", str(batch), "--volume-size", str(G)]
    env = dict(os.environ, TMPDIR="/tmp")
    with open(os.path.join(OUT, f"{TAG}pmc_{tag}.log"), "w") as log:
        rc = subprocess.run(cmd, cwd=ROOT, env=env, stdout=log, stderr=subprocess.STDOUT).returncode
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if rc != 0 or not files:
        print(f"pass {tag}: rc {rc}, csv {files}", file=sys.stderr)
        return None
    rows = list(csv.DictReader(open(files[0])))
    shutil.rmtree(d, ignore_errors=True)
    # dispatch id -> {counter: value}; a counter may be reported in several rows (per dimension): summed
    disp = collections.OrderedDict()
    for r in sorted(rows, key=lambda r: int(r["Dispatch_Id"])):
        e = disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "c": collections.defaultdict(float),
                                                    "vgpr": int(r["VGPR_Count"]) + int(r.get("Accum_VGPR_Count") or 0),
                                                    "lds": int(r["LDS_Block_Size"]), "grid": int(r["Grid_Size"]),
                                                    "wg": int(r["Workgroup_Size"]),
                                                    "us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
        e["c"][r["Counter_Name"]] += float(r["Counter_Value"])
    # cut into forwards; keep the last one
    steps, cur = [], None
    for e in disp.values():
        n = e["name"]
        if "::gather_" in n or "voxelize_kernel" in n:
            if cur is None or any(any(f in x["name"] for f in FRONT) for x in cur):
                cur = []
                steps.append(cur)
        if cur is not None and any(k in n for k in OURS) and "pack" not in n:
            cur.append(e)
    steps = [s for s in steps if any(any(f in x["name"] for f in FRONT) for x in s)]
    return steps[-1] if steps else None


def calibrate():
    """FETCH_SIZE / WRITE_SIZE on copies of a known size (tools/diag/copy_calib: 1 GiB read + 1 GiB written per launch); the
    binary is built here when missing.  Returns {kernel: {counter: mean per launch}}."""
    exe = os.path.join(ROOT, "tools", "diag", "copy_calib")
    if not os.path.isfile(exe):
        subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", exe + ".hip", "-o", exe], check=False)
    if not os.path.isfile(exe):
        return {}
    res = collections.defaultdict(dict)
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(OUT, TAG + "cal_" + c)
        shutil.rmtree(d, ignore_errors=True)
        subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", c, "--output-format", "csv", "-d", d, "--", exe],
                       cwd=ROOT, env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        acc = collections.defaultdict(list)
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per = collections.defaultdict(float)
            names = {}
            for r in csv.DictReader(open(f)):
                per[int(r["Dispatch_Id"])] += float(r["Counter_Value"])
                names[int(r["Dispatch_Id"])] = short(r["Kernel_Name"])
            for k, v in per.items():
                acc[names[k]].append(v)
        for k, v in acc.items():
            if k.startswith(("copy", "gather32")):
                res[k][c] = round(sum(v) / len(v), 1)
        shutil.rmtree(d, ignore_errors=True)
    return dict(res)


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    G = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    os.makedirs(OUT, exist_ok=True)
    sys.path.insert(0, ROOT)
    # fingerprint without importing torch here (the hash needs only the files)
    import hashlib
    csrc = os.path.join(ROOT, "sceneego_amd", "csrc")
    h = hashlib.sha256()
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h", ".sh")))
    files.append(os.path.join(ROOT, "include", "sceneego_hip.h"))
    for p in files:
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
    fingerprint = h.hexdigest()[:16]
    try:
        built = open(os.path.join(ROOT, "sceneego_amd", "libsceneego_hip.srchash")).read().strip()
    except OSError:
        built = None
    if built != fingerprint:
        print(f"libsceneego_hip.so was built from csrc {built}, the checkout is csrc {fingerprint}: rebuild before taking counters", file=sys.stderr)
        sys.exit(1)
    abi = int(re.search(r"^ABI_VERSION = (\d+)", open(os.path.join(ROOT, "sceneego_amd", "_lib.py")).read(), re.M).group(1))

    # launch keys in issue order, from bench.py's own HIP-event pass (no profiler attached)
    order_file = os.path.join(OUT, TAG + "_launch_order.json")
    rc = subprocess.run(["python3", "bench.py", "--steps", "3", "--warmup", "2", "--streams", "1", "--no-cpu-baseline", "--no-parity",
                         "--no-extras", "--batch", str(batch), "--volume-size", str(G), "--dump-launch-order", order_file],
                        cwd=ROOT, stdout=open(os.path.join(OUT, TAG + "_order_bench.log"), "w"), stderr=subprocess.STDOUT).returncode
    order = json.load(open(order_file)) if rc == 0 and os.path.isfile(order_file) else []
    w2d_keys = [k for k in order if k[0] == "conv3d" and k[1] == 3 and k[-1] in (2, 3)]      # the 2-D Winograd family: F(4,3)xF(2,3) = 2, F(4,3)xF(4,3) = 3

    table = None
    for counters in PASSES:
        step = run_pass(counters, batch, G)
        if step is None:
            continue
        if table is None:
            table = [{"pos": i, "kernel": short(e["name"]), "vgpr": e["vgpr"], "lds": e["lds"], "workgroups": e["grid"] // max(1, e["wg"]),
                      "counters": {}, "us_profiled": {}} for i, e in enumerate(step)]
        if len(step) != len(table):
            print(f"pass {counters[0]}: {len(step)} dispatches per step, expected {len(table)}", file=sys.stderr)
            continue
        for t, e in zip(table, step):
            assert t["kernel"] == short(e["name"]), (t["kernel"], e["name"])
            t["counters"].update({k: v for k, v in e["c"].items()})
            t["us_profiled"][counters[0]] = round(e["us"], 1)
    if not table:
        print("no pass produced data", file=sys.stderr)
        sys.exit(1)
    # label the 2-D Winograd and 7^3 dispatches with their shape
    it = iter(w2d_keys)
    is_w2d = lambda t: "wino2d" in t["kernel"] or "wino44pp" in t["kernel"]
    n_w2d = sum(1 for t in table if is_w2d(t))
    for t in table:
        t["shape"] = ""
        if is_w2d(t) and n_w2d == len(w2d_keys):
            k = next(it)
            t["shape"] = f"{k[2]}->{k[3]}@{k[4]}^3"
        elif "conv3d_k7" in t["kernel"]:
            t["shape"] = f"7^3 33->16@{G}^3"
    names = sorted({c for t in table for c in t["counters"]})
    lines = [f"# rocprofv3 --pmc passes over one bench.py step (B={batch}, {G}^3, fp32, --streams 1), per dispatch; ABI {abi}, csrc {fingerprint}",
             "# FETCH_SIZE / WRITE_SIZE in KiB as reported (FETCH_SIZE x 2 = bytes read on gfx950); hbm_MB = (2*FETCH + WRITE) * 1024 / 1e6",
             f"{'pos':>3s} {'kernel':48s} {'shape':16s} {'regs':>4s} {'hbm_MB':>9s} " + " ".join(f"{n[:24]:>24s}" for n in names)]
    for t in table:
        c = t["counters"]
        hbm = (2 * c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0)) * 1024 / 1e6
        lines.append(f"{t['pos']:3d} {t['kernel']:48s} {t['shape']:16s} {t['vgpr']:4d} {hbm:9.1f} " +
                     " ".join(f"{c.get(n, float('nan')):24.1f}" for n in names))
    with open(os.path.join(OUT, TAG + "_pmc_table.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))

    dom = [t for t in table if t["shape"] == f"32->32@{G}^3"]
    dom_algo = 3 if sum("wino44pp" in t["kernel"] for t in dom) * 2 > len(dom) else 2
    k7 = [t for t in table if "conv3d_k7" in t["kernel"]]
    rec = {"source": "tools/pmc_round.py on the GPU box: separate rocprofv3 --kernel-trace --pmc passes over one bench.py step, per dispatch "
                     "(profiles/<TAG>_pmc_table.txt)",
           "abi_version": abi, "csrc_sha256_16": fingerprint, "batch": batch, "volume_size": G, "algo": dom_algo,
           "kernel": f"3x3x3 32->32 @{G}^3, B={batch}: the {len(dom)} launches of this shape in one step (" +
                     ", ".join(sorted({t["kernel"] for t in dom})) + ")",
           "correction": "FETCH_SIZE x 2 (gfx950 counts 64 B per 128-B request; tools/diag/copy_calib: 1 GiB read reports 524 298 KiB), "
                         "WRITE_SIZE as is; both in KiB",
           "algorithmic_bytes_per_launch": 4 * batch * G ** 3 * 32 * 3}
    if dom:
        mean = lambda key: sum(t["counters"].get(key, 0.0) for t in dom) / len(dom)
        rec.update({"launches_per_step": len(dom), "fetch_kib_raw": round(mean("FETCH_SIZE"), 1), "write_kib": round(mean("WRITE_SIZE"), 1),
                    "hbm_bytes_per_launch": int((2 * mean("FETCH_SIZE") + mean("WRITE_SIZE")) * 1024),
                    "write_bytes_per_launch": int(mean("WRITE_SIZE") * 1024),
                    "mfma_busy_cycles_per_launch": int(mean("SQ_VALU_MFMA_BUSY_CYCLES")),
                    "lds_bank_conflict_cycles": int(mean("SQ_LDS_BANK_CONFLICT")), "lds_active_cycles": int(mean("SQ_LDS_IDX_ACTIVE")),
                    # GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES over the 256 x 4 SIMDs
                    "gui_active_cycles_all_xcds": int(mean("GRBM_GUI_ACTIVE")),
                    "mfma_pipe_busy_frac_of_launch": round(mean("SQ_VALU_MFMA_BUSY_CYCLES") / 1024.0 / max(1.0, mean("GRBM_GUI_ACTIVE") / 8.0), 4),
                    "per_launch": [{"pos": t["pos"], "kernel": t["kernel"],
                                    "hbm_bytes": int((2 * t["counters"].get("FETCH_SIZE", 0) + t["counters"].get("WRITE_SIZE", 0)) * 1024)}
                                   for t in dom]})
    cal = calibrate()
    if cal:
        rec["calibration_kib_per_launch"] = dict(cal, note="tools/diag/copy_calib: copy16 / copy8 read and write 1 GiB = 1 048 576 KiB per "
                                                 "launch, gather32of128 touches every line of 1 GiB and writes 256 MiB")
        with open(os.path.join(OUT, TAG + "_pmc_table.txt"), "a") as f:
            f.write("# calibration (KiB per launch; 1 GiB = 1048576 KiB read and written by copy16 / copy8): " + json.dumps(cal) + "\n")
    if k7:
        c = k7[0]["counters"]
        rec["conv7"] = {"kernel": k7[0]["kernel"], "fetch_kib_raw": c.get("FETCH_SIZE"), "write_kib": c.get("WRITE_SIZE"),
                        "hbm_bytes_per_launch": int((2 * c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0)) * 1024),
                        "algorithmic_bytes_per_launch": 4 * batch * G ** 3 * (33 + 16),
                        "mfma_busy_cycles": c.get("SQ_VALU_MFMA_BUSY_CYCLES"),
                        "lds_bank_conflict_cycles": c.get("SQ_LDS_BANK_CONFLICT"), "lds_active_cycles": c.get("SQ_LDS_IDX_ACTIVE")}
    fft = [t for t in table if "fft7_" in t["kernel"]]
    if fft:       # the frequency-domain front layer: its three passes, counter bytes against the byte model of bench.py's roofline.front_layer
        m_tiles = batch * (G // 16) ** 3
        rec["front_layer_fft"] = {
            "passes": [{"kernel": t["kernel"], "fetch_kib_raw": t["counters"].get("FETCH_SIZE"), "write_kib": t["counters"].get("WRITE_SIZE"),
                        "hbm_bytes": int((2 * t["counters"].get("FETCH_SIZE", 0) + t["counters"].get("WRITE_SIZE", 0)) * 1024),
                        "mfma_busy_cycles": t["counters"].get("SQ_VALU_MFMA_BUSY_CYCLES"), "lds_bank_conflict_cycles": t["counters"].get("SQ_LDS_BANK_CONFLICT"),
                        "lds_active_cycles": t["counters"].get("SQ_LDS_IDX_ACTIVE"), "insts_vmem_rd": t["counters"].get("SQ_INSTS_VMEM_RD"),
                        "insts_vmem_wr": t["counters"].get("SQ_INSTS_VMEM_WR"), "wait_any": t["counters"].get("SQ_WAIT_ANY"),
                        "wave_cycles": t["counters"].get("SQ_WAVE_CYCLES")} for t in fft],
            "hbm_bytes_per_call": int(sum((2 * t["counters"].get("FETCH_SIZE", 0) + t["counters"].get("WRITE_SIZE", 0)) * 1024 for t in fft)),
            "model_bytes_per_call": int(4.0 * batch * G ** 3 * (33 + 16) + 2.0 * m_tiles * (33 + 16) * 7488 * 8 + 7488 * 17 * 128 * 4.0),
            "algorithmic_bytes_per_call": 4 * batch * G ** 3 * (33 + 16)}
    with open(os.path.join(OUT, TAG + "_pmc.json"), "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec)[:600])


if __name__ == "__main__":
    main()
