#!/usr/bin/env bash
# The 1 -> 8 GPU curve in one command (reference: alonet/common/pl_helpers.py:365-374 picks DDP from the device
# count).  Runs `python bench.py --gpus N` for every N of the list that the node can serve, back to back, and writes ONE json:
# the per-N lines as bench.py printed them plus frames/s and weak-scaling efficiency value(N) / (N x value(1)).
#
#   tools/scale.sh [--gpus 1,2,4,8] [--out gpurun_out/scale.json] [-- extra bench.py flags]
#
# Secondary legs (RAFT, training, panoptic, CPU baselines) are kept unless the extra flags switch them off; for a quick curve:
#   tools/scale.sh -- --no-raft --train-steps 0 --panoptic-steps 0 --fp32-steps 0 --eager-steps 0 --no-cpu-baseline --micro-reps 0
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
GPUS="1,2,4,8"
OUT="$ROOT/gpurun_out/scale.json"
EXTRA=()
while [ $# -gt 0 ]; do
    case "$1" in
        --gpus) GPUS="$2"; shift 2 ;;
        --out) OUT="$2"; shift 2 ;;
        --) shift; EXTRA=("$@"); break ;;
        *) echo "usage: $0 [--gpus 1,2,4,8] [--out file.json] [-- bench.py flags]" >&2; exit 2 ;;
    esac
done
export HSA_ENABLE_IPC_MODE_LEGACY=0   # dmabuf IPC: RCCL between the ranks of one node needs it on this driver
mkdir -p "$(dirname "$OUT")"
TMP="$(mktemp -d)"
trap 'rm -rf "$TMP"' EXIT
for n in ${GPUS//,/ }; do
    echo "[scale] --gpus $n" >&2
    # a failing N (fewer devices than asked for, a rank that dies) is recorded, not fatal: the other points still make a curve
    if python "$ROOT/bench.py" --gpus "$n" "${EXTRA[@]}" > "$TMP/out_$n.txt" 2> "$TMP/err_$n.txt"; then
        grep '^{' "$TMP/out_$n.txt" | tail -1 > "$TMP/line_$n.json" || true
    else
        echo "[scale] --gpus $n failed: $(tail -2 "$TMP/err_$n.txt" | tr '\n' ' ')" >&2
    fi
done
python - "$TMP" "$OUT" "$GPUS" <<'PY'
import json, os, sys
tmp, out, gpus = sys.argv[1], sys.argv[2], [int(g) for g in sys.argv[3].split(",")]
points, failed = [], []
for n in gpus:
    path = os.path.join(tmp, f"line_{n}.json")
    if not os.path.exists(path) or not os.path.getsize(path):
        err = os.path.join(tmp, f"err_{n}.txt")
        failed.append({"n_gpus": n, "stderr_tail": open(err).read()[-500:] if os.path.exists(err) else ""})
        continue
    line = json.loads(open(path).read())
    points.append({"n_gpus": line["n_gpus"], "value": line["value"], "unit": line["unit"], "ms_per_step": line["ms_per_step"], "line": line})
base = next((p["value"] / p["n_gpus"] for p in points if p["n_gpus"] == 1), None)
for p in points:
    p["efficiency"] = None if base is None else round(p["value"] / (p["n_gpus"] * base), 4)
rec = {"metric": points[0]["line"].get("metric") if points else None, "scaling": "weak",
       "efficiency": "value(N) / (N x value(1))", "points": points, "failed": failed}
with open(out, "w") as f:
    json.dump(rec, f, indent=1)
print(json.dumps({"points": [{k: p[k] for k in ("n_gpus", "value", "ms_per_step", "efficiency")} for p in points], "failed": [f["n_gpus"] for f in failed]}))
PY
