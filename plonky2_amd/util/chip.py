"""MI355X (gfx950, CDNA4) constants the measurements are priced against -- ONE place, each citing where it comes from.
`/opt/skills/guides/MI355X_MICROARCH.md` is the image's hardware guide ("the guide" below)."""

# the guide, "HBM / memory": 8 stacks of HBM3E, 8.0 TB/s peak (a float4 copy kernel reaches ~6.3 TB/s)
HBM_PEAK_GBS = 8000.0
HBM_COPY_ACHIEVABLE_GBS = 6300.0

# the guide, hardware model: 256 CUs in 8 XCDs, 4 SIMDs per CU, wave64, peak engine clock 2.4 GHz
NUM_CUS = 256
SIMDS_PER_CU = 4
NUM_SIMDS = NUM_CUS * SIMDS_PER_CU
PEAK_CLOCK_GHZ = 2.4

# A single-issue VALU instruction (VOP2 / VOP3, 32- or 64-bit integer: v_mad_u64_u32, v_add_co_u32, v_cndmask_b32 ...) occupies a
# SIMD's 16 lanes for 4 cycles per wave64.  The guide's "Per-instruction cycle constants" row `v_fma_f32 (wave64) = 2 cyc` is the
# FP32 spec rate (157.3 TFLOP/s = 1024 SIMDs x 2.4 GHz x 128 flop / 2 cyc), which only the PACKED form reaches: profiles/ubench.json
# measures v_pk_fma_f32 at 4.41 cycles per wave64 instruction (2 FMAs per lane: 2.2 cycles per 64 FMAs = 0.90 of spec) and plain
# v_fma_f32 at 4.40-4.73.  Integer arithmetic has no packed 64-bit form, so its nominal ceiling is one instruction per 4 cycles.
CYCLES_PER_VALU_INST_WAVE64 = 4.0
VALU_NOMINAL_GWAVE_INST_PER_S = NUM_SIMDS * PEAK_CLOCK_GHZ / CYCLES_PER_VALU_INST_WAVE64   # 614.4


# Since round 6 the Poseidon S-box products assemble their 128 bits through the multiply-add's 64-bit addend (csrc/gl_mul3.hpp mul3cg /
# mul1cg): three v_mov_b32 per product that CO-ISSUE with the multiply-adds (profiles/r06_ubench_cheap.txt: a v_mov next to v_mad_u64_u32
# costs 1.5 cycles or less, not a slot of its own; in the kernel 99 % of the 4-cycle slots are taken by the other instructions,
# profiles/r06_sbox_carryfree_ab.txt, r06_sbox_hybrid_ab.txt).  The 4-cycle-slot ceiling above therefore prices the FULL-PRICE
# instructions: all VALU instructions minus these moves, whose number is a property of the algorithm -- 8 full rounds x 12 + 22 partial
# rounds = 118 S-boxes x 4 products (x^2, x^3, x^4, x^7).
POSEIDON_SBOX_PRODUCTS_PER_PERMUTATION = (8 * 12 + 22) * 4   # 472
CO_ISSUED_MOVES_PER_SBOX_PRODUCT = 3


def box_report(torch, device_index=0):
    """what the box itself reports (for the record beside the constants; the spec peaks above are what fractions are priced on)"""
    try:
        p = torch.cuda.get_device_properties(device_index)
        rep = {"name": p.name, "compute_units": getattr(p, "multi_processor_count", None), "total_memory_GiB": round(p.total_memory / 2**30, 1),
               "gcn_arch": getattr(p, "gcnArchName", None)}
        for k in ("clock_rate", "memory_clock_rate", "memory_bus_width", "L2_cache_size"):
            if hasattr(p, k):
                rep[k] = getattr(p, k)
        # what the driver exposes about clocks (raw: the level tables of the first card; `*` marks the current level)
        import glob
        for name in ("pp_dpm_mclk", "pp_dpm_sclk"):
            for path in sorted(glob.glob("/sys/class/drm/card*/device/" + name))[:1]:
                try:
                    rep[name] = open(path).read().split("\n")[:8]
                except OSError:
                    pass
        if rep.get("memory_clock_rate") and rep.get("memory_bus_width"):
            # DDR: two transfers per memory clock; kHz * bits / 8 -> GB/s
            rep["hbm_peak_GBs_from_box"] = 2 * rep["memory_clock_rate"] * 1e3 * rep["memory_bus_width"] / 8 / 1e9
        return rep
    except Exception as ex:  # noqa: BLE001
        return {"error": repr(ex)}
