import csv, sys
t={}
rows=[]
for r in csv.reader(open(sys.argv[1])):
    if not r or r[0]=="kernel" or r[0].startswith("#"): continue
    n=r[0]; us=float(r[5]); calls=float(r[1])
    if "Im2d2Col" in n or "Col2Im" in n: k="im2col"
    elif "copy_kernel" in n: k="copy"
    elif "elementwise" in n or "SubTensor" in n or "at::native" in n: k="elementwise"
    elif n.startswith("Cijk"): k="gemm"
    elif n.startswith(("rpn_","roi_","nms_","topk","proposal","head_targets","det_loss","affine","conv3x3")): k="ours"
    elif "igemm" in n or "miopen" in n.lower() or "Conv" in n or "Winograd" in n.lower(): k="miopen"
    else: k="other"
    t.setdefault(k,[0,0]); t[k][0]+=us; t[k][1]+=calls
    rows.append((us,calls,k,n[:110]))
print({k:(round(v[0]),round(v[1])) for k,v in t.items()})
for us,calls,k,n in sorted(rows,reverse=True)[:int(sys.argv[2]) if len(sys.argv)>2 else 25]:
    print("%8.1f %6.1f %-12s %s"%(us,calls,k,n))
