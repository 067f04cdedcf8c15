cd $GRAFT_REPO_ROOT
for i in 1 2 3; do for f in 1 0; do ENC_NO_FUSE=$f timeout 300 python tools/enc_fwd.py 10 2>/dev/null | tail -1 | cut -c1-60 | sed "s/^/nofuse=$f /"; done; done
