R=$GRAFT_REPO_ROOT; cd $R
cp deeptreeattention_amd/libdta_hip.so /tmp/lib_keep.so
DTA_EXTRA_HIPCC_FLAGS=-DDTA_TICKS python -m deeptreeattention_amd.build --force --no-dev > /tmp/build_ticks.log 2>&1 || tail -5 /tmp/build_ticks.log
python tools/tailticks.py
cp /tmp/lib_keep.so deeptreeattention_amd/libdta_hip.so
