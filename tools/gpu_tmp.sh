cd $GRAFT_REPO_ROOT && timeout -k 10 300 python tools/batch_size_probe.py cornellbox.obj bvh && timeout -k 10 300 python tools/batch_size_probe.py cornellbox.obj kd
