# same-box A/B of library builds on the small-batch passes: tools/_ab_small.sh "<batch sizes>" <label=path-or-empty> ...   (empty = the tree's build)
BS=$1; shift
for rep in 1 2; do
  for spec in "$@"; do
    label=${spec%%=*}; path=${spec#*=}
    out=$(env BSG_LIB=$path timeout -k 10 200 python tools/bench_small.py $BS 2>/dev/null | tail -1)
    echo "$label rep$rep: $out"
  done
done
