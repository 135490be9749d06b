import subprocess, sys, textwrap
cases = {
"A_tree": """
import numpy as np, pcl_amd
ctx = pcl_amd.Context(0)
t = pcl_amd.KdTree(ctx); t.setInputCloud(np.random.rand(1000,3).astype(np.float32))
print(t.nearestKSearch(np.random.rand(10,3).astype(np.float32), 2)[0][0])
""",
"B_torch": """
import numpy as np, pcl_amd, torch
ctx = pcl_amd.Context(0)
x = torch.rand(1000,4).cuda()
t = pcl_amd.KdTree(ctx); t.setInputCloud(x)
print(t.nearestKSearch(x[:10], 2)[0][0])
""",
"C_icp": """
import numpy as np, pcl_amd
ctx = pcl_amd.Context(0)
tgt, src, T = pcl_amd.synth.icp_pair(5000)
icp = pcl_amd.IterativeClosestPoint(ctx); icp.setInputTarget(tgt); icp.setInputSource(src); icp.setMaximumIterations(3); icp.align()
print(icp.nr_iterations_)
""",
"D_vg": """
import numpy as np, pcl_amd
ctx = pcl_amd.Context(0)
tgt, src, T = pcl_amd.synth.icp_pair(5000)
vg = pcl_amd.VoxelGrid(ctx); vg.setInputCloud(tgt); vg.setLeafSize(0.05); print(len(vg.filter()))
""",
"E_normals": """
import numpy as np, pcl_amd
ctx = pcl_amd.Context(0)
tgt, src, T = pcl_amd.synth.icp_pair(5000)
ne = pcl_amd.NormalEstimation(ctx); ne.setInputCloud(tgt); ne.setKSearch(8); print(ne.compute()[0])
""",
"F_ctx_first": """
import numpy as np, pcl_amd
ctx = pcl_amd.Context(0)
t = pcl_amd.KdTree(ctx); t.setInputCloud(np.random.rand(1000,3).astype(np.float32))
ctx.close()
del t
print('ok')
""",
"G_err": """
import numpy as np, pcl_amd
ctx = pcl_amd.Context(0)
pts = np.random.default_rng(4).uniform(-1000, 1000, (100, 3)).astype(np.float32)
vg = pcl_amd.VoxelGrid(ctx); vg.setInputCloud(pts); vg.setLeafSize(1e-4)
try:
    vg.filter()
except pcl_amd.PclHipError as e: print(e)
""",
}
for name, code in cases.items():
    r = subprocess.run([sys.executable, "-X", "faulthandler", "-c", textwrap.dedent(code)], capture_output=True, text=True, cwd=".")
    print(name, "rc=", r.returncode, r.stdout.strip()[-80:], "|", r.stderr.strip()[-300:].replace("\n"," / "))
