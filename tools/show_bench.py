import json,sys
d=json.load(open(sys.argv[1]))
def show(k,v,ind=0):
    if isinstance(v,dict):
        print(" "*ind+k+":")
        for kk,vv in v.items(): show(kk,vv,ind+2)
    else:
        sv=str(v); print(" "*ind+f"{k}: {sv[:110]}")
for k,v in d.items(): show(k,v)
