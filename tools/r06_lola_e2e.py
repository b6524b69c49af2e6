import sys, json, time
sys.path.insert(0, '.')
import numpy as np, torch
import bench
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
out = bench.bench_lola_config4(dev, stream, cpu=False)
print(json.dumps(out))
