import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "contextaware-poseformer_amd"))
from capf import lib as capf
torch.manual_seed(0)
for (B,H,W,C,res,act) in [(1,16,16,32,False,0),(1,16,16,16,True,0),(1,16,16,16,False,1),(2,16,16,16,False,0),(1,32,32,16,False,0),(1,64,64,16,False,0),(3,64,64,32,True,1)]:
    x=torch.randn(B,H,W,C).cuda()
    w=torch.randn(C,C,3,3)/(C*9)**.5
    wp,bias=capf.pack_conv_f32h2(w.cuda(),None)
    y1,=capf.conv_nhwc_f32h2_group([(x,wp,bias,1,None,C)])
    y1p,ex=capf.conv_nhwc_f32h2_planes(x,wp,bias,1,None,C,planes_out=True)
    dec=capf.planes_to_fp32(y1p,ex,capf.f32h2_tile_pixels(B,H,W))
    y2,=capf.conv_nhwc_f32h2_group([(y1,wp,bias,act,x if res else None,C)])
    y2p=capf.conv_nhwc_f32h2_planes(y1p,wp,bias,act,x if res else None,C,exps_in=ex)
    d=(y2p-y2).abs()
    bad=(d>1e-4*y2.abs().max()).nonzero()
    print((B,H,W,C,res,act),"exps",ex.shape,ex.flatten()[:8].tolist(),"decode err",(dec-y1).abs().max().item(),"max diff",d.max().item(),"ref max",y2.abs().max().item(),"bad",bad.shape[0], bad[:3].tolist())
