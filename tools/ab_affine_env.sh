#!/bin/bash
# GPU box: cfg 2 flow time under environment switches of the resident kernel (BGK_AFF_RW, BGK_AFF_NOPF), same box
for spec in "$@"; do
  echo "== $spec"; env $spec python tools/ab_affine.py 2>&1 | grep -E "variant 2|max" | tail -2
done
