#!/usr/bin/env python3
"""profiles/pmc_traffic.json (what bench.py reports as roofline.traffic) from the calibrated summary of tools/pmc_summary.py traffic.

usage: make_pmc_traffic.py <pmc_traffic_calibrated.json> <curve> <log2_domain> <source note>  > profiles/pmc_traffic.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from tree_hash import kernel_source_sha16  # noqa: E402

d = json.load(open(sys.argv[1]))
curve, log2, note = sys.argv[2], int(sys.argv[3]), sys.argv[4]
K = d["kernels"]


def pick(sub):
    for k, v in K.items():
        if sub in k:
            return k, v
    raise SystemExit("kernel not found: " + sub)


g1n, g1 = pick("bucket_accumulate30_kernel<Fp30")
g2n, g2 = pick("bucket_accumulate30_kernel<Fp2p30")
# proofs in the profiled run: one un-permute of h per proof (the pointwise quotient is fused into the seventh transform since round 4)
proofs = max(v["launches"] for k, v in K.items() if "bitrev_scale_kernel" in k)
ntt = sum(v["hbm_bytes_per_launch"] * v["launches"] for k, v in K.items() if "ntt30_" in k) / proofs
ntt_raw = sum((v["fetch_raw_bytes"] + v["write_raw_bytes"]) * v["launches"] for k, v in K.items() if "ntt30_" in k) / proofs
out = {
    "workload": {"curve": curve, "log2_domain": log2, "n_gpus": 1},
    "kernel": g1n,
    "hbm_bytes_per_launch": g1["hbm_bytes_per_launch"],
    "fetch_bytes_per_launch": g1["fetch_bytes"], "write_bytes_per_launch": g1["write_bytes"],
    "raw_counter_bytes_per_launch": {"FETCH_SIZE x 1024": g1["fetch_raw_bytes"], "WRITE_SIZE x 1024": g1["write_raw_bytes"]},
    "calibration": {"fetch": {"pattern": g1["fetch_pattern"], "factor": g1["fetch_factor"]},
                    "write": {"pattern": g1["write_pattern"], "factor": g1["write_factor"]},
                    "note": "factor = true bytes / (counter x 1024) of the known-bytes kernel with the same access pattern (tools/calib.hip), "
                            "run under the same --pmc configuration right before the bench on the same box; all factors in `calibration_kernels`"},
    "g2_bucket_pass": {"kernel": g2n, "hbm_bytes_per_launch": g2["hbm_bytes_per_launch"], "fetch_factor": g2["fetch_factor"],
                       "write_factor": g2["write_factor"], "raw_bytes": g2["fetch_raw_bytes"] + g2["write_raw_bytes"]},
    "ntt_hbm_bytes_per_step": ntt,
    "ntt_raw_counter_bytes_per_step": ntt_raw,
    "ntt_note": f"sum over the ntt30_* launches of one proof ({proofs} proofs in the pass), FETCH_SIZE / WRITE_SIZE corrected with the 32-byte "
                f"coalesced read / write factors; algorithmic 7 * 2 * 32 * n = {7 * 64 * (1 << log2) / 1e9:.2f} GB",
    "calibration_kernels": d["calibration"],
    "source": note,
    # the tree these counters belong to: bench.py reports them only when the running tree hashes the same (tools/tree_hash.py)
    "kernel_source_sha16": kernel_source_sha16(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))),
    "git_commit": os.environ.get("G16_GIT_COMMIT", "?"),   # the GPU box has no .git: passed in by the caller (tools/r05_evidence.sh)
    "g1_launch_note": "per ACTUAL launch of the G1 kernel, averaged over the launches of the profiled run (since round 5 the MSMs that are "
                      "ready together share one launch: l + a + b_g1, then h)",
}
print(json.dumps(out, indent=1))
