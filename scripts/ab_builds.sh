#!/bin/bash
# A/B of library builds on ONE box: alternates the trees, one process per sample.  usage: scripts/ab_builds.sh out.jsonl rounds tree:label ...
out=$1; rounds=$2; shift 2
: > "$out"
for r in $(seq 1 "$rounds"); do
  for tl in "$@"; do
    t=${tl%%:*}; l=${tl##*:}
    timeout 600 python scripts/ab_builds.py "$t" "$l" 60 | tail -1 >> "$out" || echo "{\"label\": \"$l\", \"failed\": true}" >> "$out"
  done
done
