import sys, os, tempfile, numpy as np, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import portal_amd as pa
from tests import synthetic
from tests.test_glsl_fuzz import fuzz_scene, N_EXPR
from oracle import host_build as hb
seed=105219
text,exprs=fuzz_scene(seed)
w,h=4*N_EXPR,12
def build(keep):
    body=["float x = hit.u;","float y = hit.v;","vec3 p = vec3(x * 1.3 - 0.2, y + 0.35, x * y + 0.6);","vec4 q = vec4(y, -x, 0.4, x - y);",f"int band = int(floor((x * 0.5 + 0.5) * {N_EXPR}.0));","vec3 c = vec3(0.0);"]
    first=True
    for k,e in enumerate(exprs):
        if k in keep:
            body.append(("if" if first else "else if")+f" (band == {k}) {{ c = {e}; }}"); first=False
    body.append("return material_simple(hit, r, abs(c) * 0.25, 0.0, false, 1.0, 0.0);")
    code="\n".join(body)
    mat=f'(name: "fuzz", data: Complex(code: (("{code}")))),'
    return synthetic.wall_scene(r=1.0,size=1.0,extra_materials=mat).replace("return wall_M; }","return fuzz_M; }")
def run(keep, flags):
    d=tempfile.mkdtemp(); path=os.path.join(d,'f.ron'); open(path,'w').write(build(keep))
    s=pa.Scene.from_file(path)
    os.environ["PTL_HIPRTC_FLAGS"]=flags
    r=pa.SceneRenderer(s,device=0); r.set_option("render_depth",2); r.set_option("view_angle",1.5)
    got=r.draw(w,h,rgba32f=True)["rgba32f"]
    os.environ["PTL_HIPRTC_FLAGS"]=""
    rr=pa.SceneRenderer(s,device=-1); rr.set_option("render_depth",2); rr.set_option("view_angle",1.5)
    hg=hb.host_kernel_for(rr,s,w,h).render(w,h)["rgba32f"]
    same=((got.view(np.uint32)==hg.view(np.uint32))|(np.isnan(got)&np.isnan(hg)))
    return int((~same.all(axis=2)).sum()), r
import subprocess, json
if len(sys.argv)>1:
    n,_=run({11,12,13}, sys.argv[1]); print(json.dumps({"flags":sys.argv[1],"bad":n})); sys.exit(0)
cands=["-mllvm -vgpr-regalloc=basic", "-mllvm -vgpr-regalloc=fast", "-mllvm -sgpr-regalloc=basic", "-mllvm -disable-machine-cse", "-mllvm -amdgpu-dce-in-ra=0", "-mllvm -disable-machine-dce",
       "-mllvm -amdgpu-sdwa-peephole=0", "-mllvm -amdgpu-dpp-combine=0", "-mllvm -amdgpu-late-codegenprepare=0", "-mllvm -amdgpu-codegenprepare-widen-constant-loads=0",
       "-mllvm -amdgpu-sched-strategy=iterative-maxocc", "-mllvm -amdgpu-sched-strategy=max-memory-clause", "-mllvm -amdgpu-si-insert-hard-clauses=0", "-mllvm -amdgpu-load-store-vectorizer=0",
       "-mllvm -amdgpu-early-inline-all=0", "-mllvm -split-spill-mode=size", "-mllvm -rematerialization=0" , "-mllvm -disable-rematerialization", "-mllvm -amdgpu-prealloc-sgpr-spill-vgprs=1",
       "-mllvm -amdgpu-unsafe-fp-atomics", "-mllvm -verify-machineinstrs"]
for f in cands:
    out=subprocess.run([sys.executable, os.path.abspath(__file__), f], capture_output=True, text=True)
    line=[l for l in out.stdout.splitlines() if l.startswith("{")]
    print(line[-1] if line else json.dumps({"flags":f,"error":(out.stderr or out.stdout)[-200:]}), flush=True)
