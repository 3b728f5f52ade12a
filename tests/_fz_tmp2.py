import sys, os, tempfile, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import portal_amd as pa
from tests import synthetic
from oracle import host_build as hb
N=48
def scene(expr):
    body=["float x = hit.u;","float y = hit.v;","vec3 p = vec3(x * 1.3 - 0.2, y + 0.35, x * y + 0.6);","vec4 q = vec4(y, -x, 0.4, x - y);",f"vec3 c = {expr};","return material_simple(hit, r, abs(c) * 0.25, 0.0, false, 1.0, 0.0);"]
    code="\n".join(body)
    mat=f'(name: "fuzz", data: Complex(code: (("{code}")))),'
    return synthetic.wall_scene(r=1.0,size=1.0,extra_materials=mat).replace("return wall_M; }","return fuzz_M; }")
w,h=4*N,12
for expr in ('vec3(p.z, q.w, (p.z < q.w) ? 1.0 : 0.0)', 'vec3(inversesqrt(abs(q.w) + 0.2), ceil(clamp(0.5, -0.25, x + 1.0)), ((p.z < q.w) ? .75 : inversesqrt(abs(q.w) + 0.2)))', 'vec3((!(((p.z < q.w) ? .75 : inversesqrt(abs(q.w) + 0.2)) <= ceil(clamp(0.5, -0.25, x + 1.0)))) ? 1.0 : 0.0, (((p.z < q.w) ? .75 : inversesqrt(abs(q.w) + 0.2)) > 0.2) ? 1.0 : 0.0, ((!(((p.z < q.w) ? .75 : inversesqrt(abs(q.w) + 0.2)) <= ceil(clamp(0.5, -0.25, x + 1.0))) || (((p.z < q.w) ? .75 : inversesqrt(abs(q.w) + 0.2)) > 0.2))) ? 1.0 : 0.0)', 'reflect(((q.xyz / (x + 2.5)) / ((7.5 * -1.5) + 2.5)), normalize(q.xyz + vec3(0.1, 0.7, 0.2)))', '(-(p))', '((!(((p.z < q.w) ? .75 : inversesqrt(abs(q.w) + 0.2)) <= ceil(clamp(0.5, -0.25, x + 1.0))) || (((p.z < q.w) ? .75 : inversesqrt(abs(q.w) + 0.2)) > 0.2)) ? (-(p)) : reflect(((q.xyz / (x + 2.5)) / ((7.5 * -1.5) + 2.5)), normalize(q.xyz + vec3(0.1, 0.7, 0.2))))'):
    d=tempfile.mkdtemp(); path=os.path.join(d,'f.ron'); open(path,'w').write(scene(expr))
    s=pa.Scene.from_file(path)
    r=pa.SceneRenderer(s,device=0); r.set_option("render_depth",2); r.set_option("view_angle",1.5)
    got=r.draw(w,h,rgba32f=True)["rgba32f"]
    rr=pa.SceneRenderer(s,device=-1); rr.set_option("render_depth",2); rr.set_option("view_angle",1.5)
    hg=hb.host_kernel_for(rr,s,w,h).render(w,h)["rgba32f"]
    same=((got.view(np.uint32)==hg.view(np.uint32))|(np.isnan(got)&np.isnan(hg)))
    bad=np.argwhere(~same.all(axis=2))
    print(expr, "bad", len(bad))
    for y,x in bad[:4]: print("  px",x,y,"gpu",got[y,x,:3].view(np.uint32),got[y,x,:3],"host",hg[y,x,:3].view(np.uint32),hg[y,x,:3])
    print("  px 92,0: gpu", got[0,92,:3].view(np.uint32), "host", hg[0,92,:3].view(np.uint32), got[0,92,:3])
