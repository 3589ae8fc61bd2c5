cd $GRAFT_REPO_ROOT
python - <<'PY'
import json, bench
r = bench.end_to_end("C2")
print({k: (v.get('median_ms'), v.get('frames_per_s'), v.get('nodes_last'), v.get('cost_first'), v.get('cost_last')) for k, v in r.items() if isinstance(v, dict)})
PY
