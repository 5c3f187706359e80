import os, subprocess, sys, tempfile, time
ROOT="/root/repo"
sys.path.insert(0, os.path.join(ROOT,"scripts")); sys.path.insert(0, os.path.join(ROOT,"plugins"))
from config1_e2e import env_for, GST
env=env_for(tempfile.mkdtemp()); env["GSTAMD_ELEMENT_STATS"]="1"
for extra in ({}, {"GSTAMD_NO_PINNED_POOLS":"1"}):
    e=dict(env, **extra)
    for i in range(2):
        t0=time.perf_counter()
        r=subprocess.run([GST,"-q"]+"videotestsrc num-buffers=300 pattern=smpte ! video/x-raw,format=NV12,width=1920,height=1080,framerate=300/1 ! amdvideoconvertscale ! video/x-raw,format=BGRA ! fakesink sync=false".split(), env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
        dt=time.perf_counter()-t0
    print(extra, round(dt,3), r.stdout.strip()[-400:])
