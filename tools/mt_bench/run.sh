#!/bin/bash
# on the GPU box: every built variant on the three shapes
cd "$(dirname "$0")"
for b in "$@"; do
  echo "== $b"
  ./$b 4096 50000 3 0; ./$b 4096 20000 6 0; ./$b 1 100000 2 0
done
