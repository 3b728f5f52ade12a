import sys, os, tempfile, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import portal_amd as pa
from tests.test_glsl_fuzz import fuzz_scene, N_EXPR
from oracle.portal_oracle import Oracle
seed=105219
text,exprs=fuzz_scene(seed); d=tempfile.mkdtemp(); path=os.path.join(d,'f.ron'); open(path,'w').write(text)
w,h=4*N_EXPR,12
o=Oracle(path); o.options.update(render_depth=2,view_angle=1.5)
want=o.render(w,h)["rgba32f"]
for flags,env in ((0,""),(0,"-DPTL_PLAIN_SQRT"),(0,"-DPTL_PLAIN_RCP"),(0,"-DPTL_PLAIN_SQRT_RCP")):
    os.environ["PTL_HIPRTC_FLAGS"]=env
    r=pa.SceneRenderer(pa.Scene.from_file(path),device=0,flags=flags); r.set_option("render_depth",2); r.set_option("view_angle",1.5)
    got=r.draw(w,h,rgba32f=True)["rgba32f"]
    import hashlib; print("code", hashlib.sha1(r.code_object()).hexdigest()[:10], len(r.code_object()), r.resources())
    same=((got.view(np.uint32)==want.view(np.uint32))|(np.isnan(got)&np.isnan(want)))
    bad=np.argwhere(~same.all(axis=2))
    print("flags",env,"bad pixels",len(bad))
    for y,x in bad[:3]:
        print(" px",x,y,"expr",x//4, "gpu",got[y,x], got[y,x].view(np.uint32), "oracle",want[y,x].view(np.uint32))

from oracle import host_build as hb
sc=pa.Scene.from_file(path); rr=pa.SceneRenderer(sc,device=-1); rr.set_option("render_depth",2); rr.set_option("view_angle",1.5)
hg=hb.host_kernel_for(rr,sc,w,h).render(w,h)["rgba32f"]
same=((hg.view(np.uint32)==want.view(np.uint32))|(np.isnan(hg)&np.isnan(want)))
print("host build vs oracle ON THIS BOX: bad", int((~same.all(axis=2)).sum()))
print("host px92", hg[0,92].view(np.uint32), hg[1,92].view(np.uint32))
import platform, subprocess
print(subprocess.run("lscpu | grep -i 'model name'", shell=True, capture_output=True, text=True).stdout.strip())
