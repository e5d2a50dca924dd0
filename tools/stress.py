import sys, time, numpy as np
sys.path.insert(0,'.')
import __graft_entry__ as e
pkg=e.load_package(); r=pkg.HipRenderer(0)
def run(name,w,h,spp,b,p,**kw):
    sc=pkg.scene_by_name(name); r.upload_scene(sc); cam=pkg.cornell_camera(w,h)
    rp=pkg.RenderParams(spp=spp,min_bounces=b,absorb=p,seed=3,**kw)
    t=time.time(); img,g,st=r.render(cam,rp,backward=True); dt=time.time()-t
    print(f"{name} {w}x{h}x{spp} b{b} p{p} {kw}: {st['segments']/1e6:.1f} Mseg {dt*1e3:.1f} ms batches {st['batches']} finite {np.isfinite(img).all() and np.isfinite(g).all()} mean {img.mean():.5f} gmax {np.abs(g).max():.4g}", flush=True)
    return img,g
run("cornell",4096,4096,4,8,1.0)
run("cornell",64,64,4096,8,1.0)
run("cornell",8192,2,3,2,0.05)
run("cornell",3,8191,3,1,0.05)
run("cornell_specular",1000,700,7,1,0.02)          # very long roulette paths under the cap of 64
run("cornell_mirror_wall",777,333,5,1,0.05)
a=run("cornell",1024,1024,16,4,0.3,batch_paths=1<<20)
b=run("cornell",1024,1024,16,4,0.3)
print("batch independence img", np.abs(a[0]-b[0]).max(), "grad rel", np.abs(a[1]-b[1]).max()/np.abs(b[1]).max())
imgs=[]; gs=0
for k in range(7):
    i,g=run("mesh40x40",300,211,4,3,0.3,shard=k,n_shards=7,band_rows=5); imgs.append(i); gs=gs+g
full=run("mesh40x40",300,211,4,3,0.3)
print("7 ragged shards tile the frame: img", np.abs(sum(imgs)-full[0]).max(), "grad rel", np.abs(gs-full[1]).max()/np.abs(full[1]).max())
