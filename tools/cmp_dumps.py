import numpy as np, sys
a=np.load(sys.argv[1]); b=np.load(sys.argv[2])
for k in a.files:
    if k.startswith("grad_") or k in ("loss","logits"):
        x,y=a[k].astype(np.float64),b[k].astype(np.float64)
        print(k, "max|ref|", float(np.abs(y).max()), "max err", float(np.abs(x-y).max()), "rel", float(np.abs(x-y).max()/max(np.abs(y).max(),1e-30)))
