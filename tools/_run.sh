cd $GRAFT_REPO_ROOT
python - <<'PY'
import json, bench
r = bench.end_to_end("C2")
print(json.dumps(r, indent=1))
PY
python - <<'PY'
import numpy as np, subprocess, os
from dynfu_amd import synth, build as B
cfg = synth.CONFIGS["C2"]
np.stack([synth.depth_frame(cfg, f) for f in range(8)]).astype("<u2").tofile("/tmp/f.u16")
env = dict(os.environ, DFA_HOST_PROFILE="1")
r = subprocess.run([B.SEQ_BENCH, "/tmp/f.u16", "640", "480", "8", "512", "northstar"], capture_output=True, text=True, env=env)
print(r.stderr[-2500:])
PY
