"""CPU numerics probe (analysis only, imports the oracle): what happens to DEFT outputs if every conv is evaluated as a sum of
bf16 x bf16 partial products (3 / 6 terms) with fp32 accumulation, i.e. fp32 emulated on the bf16 MFMA path.  See DESIGN.md section 8."""
import sys, torch, torch.nn.functional as F
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
import deft_oracle as O
torch.set_grad_enabled(False)
orig = F.conv2d
def split(t, n):
    parts=[]; r=t
    for _ in range(n):
        p=r.bfloat16().float(); parts.append(p); r=r-p
    return parts
def make(terms):
    # terms: list of (i,j) index pairs of x-part i and w-part j
    n = max(max(i,j) for i,j in terms)+1
    def conv(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
        xs=split(x,n); ws=split(w,n)
        y=None
        for i,j in terms:
            t=orig(xs[i], ws[j], None, stride, padding, dilation, groups)
            y = t if y is None else y+t
        if b is not None: y = y + b.view(1,-1,1,1)
        return y
    return conv
T3=[(0,0),(0,1),(1,0)]
T6=[(0,0),(0,1),(1,0),(0,2),(2,0),(1,1)]
for (H,W) in [(224,384),(608,1088)]:
    sd=O.synth_state_dict("mot")
    x=torch.randn(1,3,H,W,generator=torch.Generator().manual_seed(0))
    F.conv2d=orig
    out,maps=O.dlaseg_forward(x,sd,"mot"); ref=O.generic_decode(O.sigmoid_output(out),K=100)
    gaps=(ref["scores"][0,:-1]-ref["scores"][0,1:])
    print(H,W,"min adjacent score gap %.2e, median %.2e"%(float(gaps.min()), float(gaps.median())))
    for name,terms in (("bf16x3",T3),("bf16x6",T6)):
        F.conv2d=make(terms)
        o2,m2=O.dlaseg_forward(x,sd,"mot"); d2=O.generic_decode(O.sigmoid_output(o2),K=100)
        F.conv2d=orig
        same=bool(torch.equal(d2["inds"],ref["inds"]))
        nd=int((d2["inds"]!=ref["inds"]).sum())
        print("  %s: inds equal %s (%d differ)  max|dscore| %.2e  max|dbbox| %.2e  max rel fmap err %.2e" % (name, same, nd,
              float((d2["scores"]-ref["scores"]).abs().max()), float((d2["bboxes"]-ref["bboxes"]).abs().max()),
              max(float((a-b).abs().max()/b.abs().max()) for a,b in zip(m2,maps))))
