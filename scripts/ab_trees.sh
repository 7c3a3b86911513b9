# round-over-round on ONE box: the default bench (all three arithmetics in one run) from another checkout of the repository (e.g. the
# previous round's tree unpacked and built under build_variants/) and from this tree, alternately, inside one gpurun call
#   bash scripts/ab_trees.sh build_variants/r05tree
OLD=$1
for rep in 1 2 3; do
  for tree in $OLD .; do
    (cd $tree && python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
o = d['extra']['other_prec']
print('$tree', 'fp32 %.2f ms (unit %.2f)' % (d['ms_per_step'], d['roofline']['path']['ms']), ' '.join('%s %.2f (unit %.2f)' % (k, v['ms_per_step'], v['tnet_fwd_bwd_ms']) for k, v in o.items()))")
  done
done
