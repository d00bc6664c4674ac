"""A/B of pricing-kernel builds / grid sizes on the 1M-node / 16M-arc sweep (ms per launch)."""
import os, subprocess, sys, json
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, str(ROOT))
    from network_flow_solver_amd import engine, generators
    inst = generators.named_instance(os.environ.get("WL", "netgen_1m_16m"))
    out = {}
    for pb in [int(x) for x in os.environ.get("GRIDS", "512,1024,2048").split(",")]:
        with engine.McfEngine(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=0, price_blocks=pb) as eng:
            eng.solve(max_pivots=200)     # realistic state: some arcs basic, potentials moved
            ts = [eng.time_pricing(reps=20) for _ in range(3)]
            out[pb] = round(min(ts) * 1e3, 1)
    print(json.dumps(out))
else:
    for lib in sorted((ROOT / "network_flow_solver_amd/csrc/variants").glob("*.so")):
        env = dict(os.environ, MCF_HIP_LIB=str(lib))
        r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        print(lib.name, r.stdout.strip(), r.stderr.strip()[-300:] if r.returncode else "", flush=True)
