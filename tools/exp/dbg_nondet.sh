cd $GRAFT_REPO_ROOT
for ph in init fwd bwd; do
python3 tools/exp/dbg_nondet.py $ph > /tmp/a.txt 2>/dev/null; python3 tools/exp/dbg_nondet.py $ph > /tmp/b.txt 2>/dev/null
echo "phase=$ph"; diff /tmp/a.txt /tmp/b.txt | head -12
done
