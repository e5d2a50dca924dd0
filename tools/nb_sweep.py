import sys, time, os, numpy as np
sys.path.insert(0,'.')
import __graft_entry__ as e
pkg=e.load_package(); r=pkg.HipRenderer(0)
name,b,p=sys.argv[1],int(sys.argv[2]),float(sys.argv[3])
sc=pkg.scene_by_name(name); r.upload_scene(sc); cam=pkg.cornell_camera(512,512)
rp=pkg.RenderParams(spp=64,min_bounces=b,absorb=p,seed=3)
for _ in range(3): img,g,st=r.render(cam,rp,backward=True)
t=time.time()
for _ in range(10): img,g,st=r.render(cam,rp,backward=True)
dt=(time.time()-t)/10
print(f"NB={os.environ.get('DRT_HIP_SHADE_BOUNCES','auto')} {name} b{b} p{p}: {st['segments']/1e6:.1f} Mseg {dt*1e3:.2f} ms {st['segments']/dt*1e-9:.1f} Gray/s gsum {np.abs(g).sum():.6g}")
