for abl in 0 1 2 4 8 3 15; do echo -n "abl=$abl  "; DN_FUSE_ABL=$abl timeout 120 python tools/fuse_ab.py 2>&1 | tail -1; done
for wv in 1 4; do echo -n "waves=$wv  "; DN_FUSE_MLP_WAVES=$wv timeout 120 python tools/fuse_ab.py 2>&1 | tail -1; done
