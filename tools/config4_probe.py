import os,sys,time
sys.path.insert(0,'/root/repo')
import torch
from benchkit.config4 import config4
dev=torch.device("cuda",0); torch.cuda.set_device(0)
for rep in range(2):
    r=config4(dev,0,1,dev,8,batched=True)
    print("batched", round(r["ms_per_step"],3), r["units_gathered"], r["matched_keypoints_per_step"])
