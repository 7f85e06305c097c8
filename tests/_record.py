"""Measured parity figures (flip counts of integer outputs, worst errors) written next to the pass/fail verdicts of the GPU tests:
gpurun_out/parity_counts.json, copied to profiles/ for the record (DESIGN.md section 6 quotes them)."""
import json
import os

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_counts.json")


def record(key, value):
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    d = json.load(open(OUT)) if os.path.exists(OUT) else {}
    d[key] = value
    with open(OUT, "w") as f:
        json.dump(d, f, indent=1, sort_keys=True)
