import torch, subprocess
try:
    print("clock_rate", torch.cuda.clock_rate(), "temp", torch.cuda.temperature(), "power", torch.cuda.power_draw())
except Exception as e:
    print("torch probe failed:", repr(e)[:200])
try:
    print(subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True, timeout=30).stdout[-1500:])
except Exception as e:
    print("rocm-smi failed", repr(e)[:200])
