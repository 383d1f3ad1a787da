cd $GRAFT_REPO_ROOT
for mode in "" "" ""; do
  for rep in 1 2 3; do
    D=$(mktemp -d)
    for r in 0 1 2 3 4 5 6 7; do DTA_PROBE_DEBUG=$mode python deeptreeattention_amd/peer_probe.py $D $r 8 0 > $D/out$r.txt 2> $D/err$r.txt & done
    wait
    echo "mode=[$mode] rep $rep: $(grep -h -c . $D/err0.txt) lines; $(cat $D/err*.txt | grep -v "^rank" | sort | uniq -c | head -3)"
    [ -n "$mode" ] && grep -h "^rank 0" $D/err0.txt | head -6
    rm -rf $D
  done
done
