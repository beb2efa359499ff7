import time, ctypes, torch
torch.cuda.init(); x = torch.zeros(4, device="cuda"); torch.cuda.synchronize()
hip = ctypes.CDLL("libamdhip64.so")
def cost(tag):
    s = torch.cuda.current_stream()
    ts = []
    for _ in range(2000):
        t = time.perf_counter(); s.synchronize(); ts.append(time.perf_counter() - t)
    ts.sort(); print(tag, "median us", round(1e6 * ts[len(ts)//2], 2), "p90", round(1e6 * ts[int(len(ts)*0.9)], 2), flush=True)
cost("fresh")
hs = []
for i in range(64):
    h = ctypes.c_void_p(); hip.hipStreamCreateWithFlags(ctypes.byref(h), 1); hs.append(h)      # non-blocking
cost("+64 non-blocking streams")
for i in range(64):
    h = ctypes.c_void_p(); hip.hipStreamCreate(ctypes.byref(h)); hs.append(h)
cost("+64 blocking streams")
ts = [torch.cuda.Stream() for _ in range(16)]
cost("+16 torch streams")
