#!/usr/bin/env python3
"""Objectives of the BASELINE.json-size stand-ins, computed by the oracle (oracle/ref_simplex.c: the reference's
algorithm restated in C and pinned against the reference-made goldens).  The reference itself cannot run these
sizes (dense (n-1)^2 basis matrix, SURVEY.md headline fact 5).

    python tests/golden/make_baseline_objectives.py        # ~1 minute; writes baseline_objectives.json

Each entry also records the engine's own integer algorithm (CPU emulation, oracle/emul_engine.cpp) as a second,
independent computation of the same optimum.  netgen_1m_16m is far beyond both CPU paths: its entry is written by
``--record-1m OBJECTIVE`` from a GPU run whose result passed tests/conftest.py:check_optimality (a certificate of
optimality that needs no oracle) and only serves as a regression pin.
"""
import argparse
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

import oracle  # noqa: E402
from network_flow_solver_amd import generators  # noqa: E402

OUT = Path(__file__).resolve().parent / "baseline_objectives.json"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--record-1m", type=int, default=None)
    args = ap.parse_args()
    data = json.loads(OUT.read_text()) if OUT.exists() else {}
    if args.record_1m is not None:
        inst = generators.named_instance("netgen_1m_16m")
        data["netgen_1m_16m"] = {"sha256": inst.sha256(), "n": inst.n, "m": inst.m, "objective": int(args.record_1m),
                                 "source": "MI355X engine, certified optimal by check_optimality (regression pin)"}
    else:
        for name in ("netgen_8_14a", "gridgen_8_14a"):
            inst = generators.named_instance(name)
            ref = oracle.solve_soa(inst, "dantzig", reference_order=False)
            em = oracle.emul_solve(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=2)
            assert ref["status"] == em["status"] == "optimal"
            assert int(round(ref["objective"])) == em["objective"], (name, ref["objective"], em["objective"])
            data[name] = {"sha256": inst.sha256(), "n": inst.n, "m": inst.m, "objective": em["objective"],
                          "oracle_pivots_dantzig": ref["iterations"], "oracle_seconds": round(ref["seconds"], 1),
                          "source": "oracle/ref_simplex.c (Dantzig) == oracle/emul_engine.cpp (candidate list)"}
            print(name, data[name], flush=True)
    OUT.write_text(json.dumps(data, indent=1, sort_keys=True) + "\n")


if __name__ == "__main__":
    main()
