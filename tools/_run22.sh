cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in libppp_mi355x.so libppp_mw5.so libppp_mw6.so; do
for case in 128p9 140p7; do
PPP_LIB=$GRAFT_REPO_ROOT/patchperpix_amd/csrc/$lib timeout 300 python3 tools/time_s2.py --case $case --reps 3 2>/dev/null | tail -1
done
done
