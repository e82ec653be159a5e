#!/bin/bash
# round 6, closing run: what the driver runs at round end — the whole GPU suite, smoke(), the default bench line
O=gpurun_out; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -x > $O/r6z_full.txt 2>&1; echo "full suite rc=$?"; tail -4 $O/r6z_full.txt | head -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > $O/r6z_bench.json 2> $O/r6z_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6z_bench.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "value_cold", "value_cold_ramped", "value_unpruned")})
print("roofline", d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["traffic_profile"], "c3", d["c3"]["value"], "c5dir", d["c5"]["directory"]["seconds_total"])
lr = d["long_reads"]; print("long", lr["reads_5kb"]["bases_per_s"], lr["contigs_500kb"]["bases_per_s"], lr["fasta_file"]["bases_per_s"])
print("e2e", {k: (v["value"] if isinstance(v, dict) and "value" in v else None) for k, v in d["e2e"].items()})
print("errors", [k for k in d if k.endswith("_error")])
PY
