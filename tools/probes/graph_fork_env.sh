#!/bin/bash
# tools/probes/graph_fork.py under runtime knobs (one per run): does any of them change the pace at which a replayed graph's nodes start?
export FORK_QUICK=1
echo "== default"; python tools/probes/graph_fork.py 2>&1 | grep order
for kv in "$@"; do
  echo "== $kv"; env $kv timeout 120 python tools/probes/graph_fork.py 2>&1 | grep -E "order|rror" | head -3
done
