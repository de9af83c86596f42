#!/bin/bash
# A/B two builds of libiqvit.so on ONE box, alternating (boxes differ by 1-2 %, more than most kernel changes):
#   scripts/ab_step.sh <old.so> [bench args...]     -- the in-tree library is "new"
old=$1; shift
cd $GRAFT_REPO_ROOT
cp vit-vs-raw-iq_amd/libiqvit.so /tmp/ab_new.so
run() { python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for i in 1 2 3; do
  cp $old vit-vs-raw-iq_amd/libiqvit.so; echo "old $(run "$@")"
  cp /tmp/ab_new.so vit-vs-raw-iq_amd/libiqvit.so; echo "new $(run "$@")"
done
