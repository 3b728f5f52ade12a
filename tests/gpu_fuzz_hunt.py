#!/usr/bin/env python3
"""tests/gpu_fuzz_hunt.py -- a larger one-off run of the tests' differential fuzzers on the GPU (60 random scenes through the five builds in turn; a bounded slice of it runs inside the suite: tests/test_gpu_oracle_fullsize.py, and 30 x 48 random GLSL expressions): gfx950 against the numpy oracle, bit for bit.  Development aid.
usage: gpu_fuzz_hunt.py [SEED_BASE [N_SCENES [N_GLSL]]] -- e.g. `1000 240 120` was run after the sqrt / reciprocal change: no mismatch
in 240 scenes and 5 760 expressions."""
import sys, os, tempfile, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import portal_amd as pa
from tests.test_scene_fuzz import random_scene
from tests.test_glsl_fuzz import fuzz_scene, fuzz_scene_with_uniforms, N_EXPR
from oracle.portal_oracle import Oracle
def same(a,b): return ((a.view(np.uint32)==b.view(np.uint32))|(np.isnan(a)&np.isnan(b))).all()
bad=0
BASE=int(sys.argv[1]) if len(sys.argv)>1 else 0   # shift both seed ranges for another, disjoint hunt
N_SCENES=int(sys.argv[2]) if len(sys.argv)>2 else 60
N_GLSL=int(sys.argv[3]) if len(sys.argv)>3 else 30
for seed in range(300+BASE,300+BASE+N_SCENES):
    text,cam,sub=random_scene(seed); d=tempfile.mkdtemp(); path=os.path.join(d,'r.ron'); open(path,'w').write(text)
    # the builds in turn: un-specialised, clip-constant, Bool / Int baked (zero patterns of the run-time matrices compiled in), everything baked, patterns only
    flags=(0, pa.FLAG_SPECIALIZE_STATIC, pa.FLAG_SPECIALIZE_INTS, pa.FLAG_SPECIALIZE_INTS|pa.FLAG_SPECIALIZE_ALL, pa.FLAG_SPECIALIZE_PATTERNS)[seed%5]
    r=pa.SceneRenderer(pa.Scene.from_file(path),device=0,flags=flags); r.set_option("render_depth",10); r.set_option("in_subspace",1 if sub else 0)
    r.set_camera(cam["look_at"],cam["alpha"],cam["beta"],cam["r"])
    got=r.draw(40,24,rgba32f=True)["rgba32f"]
    o=Oracle(path); o.options["render_depth"]=10; o.camera=dict(cam,in_subspace=sub)
    if not same(got,o.render(40,24)["rgba32f"]): bad+=1; print("scene seed",seed,"DIFF")
for seed in range(100000+BASE,100000+BASE+N_GLSL) if BASE else range(400,430):
    # every other one with uniform leaves (glsl_hoist)
    text,_=(fuzz_scene_with_uniforms if seed%2 else fuzz_scene)(seed); d=tempfile.mkdtemp(); path=os.path.join(d,'f.ron'); open(path,'w').write(text)
    w,h=4*N_EXPR,12
    r=pa.SceneRenderer(pa.Scene.from_file(path),device=0,flags=pa.FLAG_SPECIALIZE_INTS if seed%4==1 else 0); r.set_option("render_depth",2); r.set_option("view_angle",1.5)  # (masked products every fourth)
    got=r.draw(w,h,rgba32f=True)["rgba32f"]
    o=Oracle(path); o.options.update(render_depth=2,view_angle=1.5)
    if not same(got,o.render(w,h)["rgba32f"]): bad+=1; print("glsl seed",seed,"DIFF")
print(f"hunt done (base {BASE}, {N_SCENES} scenes, {N_GLSL} x {N_EXPR} expressions): bad =",bad)
