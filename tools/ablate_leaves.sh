#!/bin/bash
# timing ablations of the leaves kernel (results invalid when a flag is set): prints leaves ms per flag set
for f in 0 1 2 3 4 7 16 18 23; do
  echo "== DNAGPU_DEBUG_LEAVES=$f"
  DNAGPU_DEBUG_LEAVES=$f python tools/phase_probe.py ${1:-1e9} 2>&1 | grep -E "k=31|leaves" | head -2
done
