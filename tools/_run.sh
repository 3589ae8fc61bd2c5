cd $GRAFT_REPO_ROOT
cp dynfu_amd/libdynfu_amd.so /tmp/prod.so
for v in prod pf2; do
[ $v = pf2 ] && cp dynfu_amd/libdynfu_amd_pf2.so.bin dynfu_amd/libdynfu_amd.so
for rc in 256 320 448; do
DFA_S6_RC=$rc DFA_TAG=$v-rc$rc python tools/ns_assemble_time.py C3 2>&1 | tail -1
done
DFA_S6_RC=320 DFA_TAG=$v-rc320 python tools/ns_assemble_time.py C2 2>&1 | tail -1
done
cp /tmp/prod.so dynfu_amd/libdynfu_amd.so
