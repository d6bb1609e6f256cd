import json
import os
import sys

for line in sys.stdin:
    if line.startswith('{"metric"'):
        d = json.loads(line)
        g = d.get("gather", {})
        print(os.environ.get("MODE"), "frame-pairs/s", round(d["value"]), "ms/step", round(d["ms_per_step"], 3), "bytes/step",
              d.get("gathered_bytes_per_step"), {k: g.get(k) for k in ("registered_segment", "dma_bytes_last_run")})
