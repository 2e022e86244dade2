import sys, importlib, json, struct
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
abi=importlib.import_module("misaki-render_amd.abi"); hm=importlib.import_module("misaki-render_amd.hostmirror")
import oracle_binding
orc=oracle_binding.load()
g=json.load(open('/root/repo/tests/golden/rgb2spec_triplets.json'))
tab={tuple(round(x,7) for x in v['rgb']): tuple(struct.unpack('>f',bytes.fromhex(h))[0] for h in v['coeff_hex']) for v in g.values()}
look=lambda rgb: tab[tuple(round(float(x),7) for x in rgb)]
fs=hm.cbox_scene(256,256,coeff_lookup=look)
ctx=abi.Context(0); gs=abi.Scene(ctx,fs); os_=orc.scene(fs)
pixels=np.array([[128,30],[100,200],[30,128],[220,128],[128,128],[90,140]],np.int32)
for md in (1,2,3,-1):
    prm=abi.render_params(spp=64, max_depth=md)
    gx,gp=gs.sample_pixels(prm,pixels); ox,op=os_.sample_pixels(prm,pixels)
    d=(gx.view(np.uint32)!=ox.view(np.uint32)).any(-1)
    print("max_depth",md,"differing samples",d.sum(),"of",d.size, "pos equal", np.array_equal(gp,op))
    idx=np.argwhere(d)[:5]
    for i,j in idx:
        print("  pix",pixels[i],"s",j,"gpu",gx[i,j],"orc",ox[i,j], "rel", np.abs(gx[i,j]-ox[i,j])/np.maximum(np.abs(ox[i,j]),1e-20))
