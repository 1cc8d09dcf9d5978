import sys, time, json
sys.path.insert(0, '.')
import numpy as np, torch
from plonky2_amd import Engine
from plonky2_amd.fri.oracle import FriBatchInfo, PolynomialBatch, eval_openings, prove_openings
from plonky2_amd.iop.challenger import Challenger
from plonky2_amd.plonk.prover import all_wires_permutation_partial_products, compute_quotient_polys
from plonky2_amd.util.synthetic import splitmix_columns_torch
P = 0xFFFFFFFF00000001
eng = Engine(0); dev = eng.mem.device
log_n, rb, cap = 20, 3, 4
n = 1 << log_n
wires = splitmix_columns_torch(torch, dev, 0, 135, n)
cs = splitmix_columns_torch(torch, dev, 1000, 84, n)
quo = splitmix_columns_torch(torch, dev, 2000, 16, n)
k_is = [pow(14293326489335486720, j, P) for j in range(80)]
b_cs = PolynomialBatch.from_values(cs, rb, False, cap, engine=eng)
b_w = PolynomialBatch.from_values(wires, rb, False, cap, engine=eng)
zs = all_wires_permutation_partial_products(wires[:80], cs[4:84], k_is, 8, [3, 5], [11, 13], eng)
b_z = PolynomialBatch.from_values(zs, rb, False, cap, engine=eng)
widths = (84, 135, 20, 16)
allp = [(oi, pi) for oi, W in enumerate(widths) for pi in range(W)]
nxt = [(2, 0), (2, 1)]
def run(b_q, label):
    for rep in range(3):
        ch = Challenger(eng); ch.observe_elements(np.arange(8, dtype=np.uint64))
        zeta = ch.get_extension_challenge(); gz = [(zeta[0] * 7) % P, zeta[1]]
        eng.profile(True); eng.profile_results(reset=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pr = prove_openings([FriBatchInfo(zeta, allp), FriBatchInfo(gz, nxt)], [b_cs, b_w, b_z, b_q], ch, rb, cap, [4, 4, 4, 4], 16, 28, engine=eng)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
        prof = eng.profile_results(reset=True); eng.profile(False)
        print(label, rep, round(dt, 2), pr["pow_witness"], {k: round(v["ms"], 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:6]})
run(PolynomialBatch.from_coeffs(quo, rb, False, cap, engine=eng), "random-device")
chunks = compute_quotient_polys(b_w, b_cs, 4, b_z, k_is, 8, [3, 5], [11, 13], [17, 19], engine=eng)
run(PolynomialBatch.from_coeffs(chunks, rb, False, cap, engine=eng), "quotient-cols")
print("---- whole path, per repetition")
def path():
    st = {}; t = [time.perf_counter()]
    def lap(l):
        torch.cuda.synchronize(); now = time.perf_counter(); st[l] = round((now - t[0]) * 1e3, 2); t[0] = now
    b_w = PolynomialBatch.from_values(wires, rb, False, cap, engine=eng); lap("wires")
    zs = all_wires_permutation_partial_products(wires[:80], cs[4:84], k_is, 8, [3, 5], [11, 13], eng); lap("pp")
    b_z = PolynomialBatch.from_values(zs, rb, False, cap, engine=eng); lap("zs")
    chunks = compute_quotient_polys(b_w, b_cs, 4, b_z, k_is, 8, [3, 5], [11, 13], [17, 19], engine=eng); lap("quot")
    b_q = PolynomialBatch.from_coeffs(chunks, rb, False, cap, engine=eng); lap("qcommit")
    ch = Challenger(eng); ch.observe_elements(np.arange(8, dtype=np.uint64))
    zeta = ch.get_extension_challenge(); gz = [(zeta[0] * 7) % P, zeta[1]]
    eval_openings([b_cs, b_w, b_z, b_q], [zeta], eng); eval_openings([b_z], [gz], eng); lap("openings")
    pr = prove_openings([FriBatchInfo(zeta, allp), FriBatchInfo(gz, nxt)], [b_cs, b_w, b_z, b_q], ch, rb, cap, [4, 4, 4, 4], 16, 28, engine=eng); lap("prove")
    st["w"] = pr["pow_witness"]
    return st
for rep in range(5):
    print(rep, path())
