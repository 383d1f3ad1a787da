R=$GRAFT_REPO_ROOT; cd $R
cp deeptreeattention_amd/libdta_hip.so /tmp/lib_keep.so
DTA_EXTRA_HIPCC_FLAGS=-DDTA_TICKS python -m deeptreeattention_amd.build --force > /tmp/build_ticks.log 2>&1 || tail -5 /tmp/build_ticks.log
echo "== D2 (default)"; python tools/wticks.py
echo "== no D2"; DTA_NO_WGRAD_D2=1 python tools/wticks.py
cp /tmp/lib_keep.so deeptreeattention_amd/libdta_hip.so
