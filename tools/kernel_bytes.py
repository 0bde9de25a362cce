"""Algorithmic bytes and flops per launch of the kernels of the batch schedule (every tensor a kernel reads or writes counted
ONCE, at the size the data layout of DESIGN.md section 2 gives it), by workload.  tools/pmc_aggregate.py holds the FETCH_SIZE /
WRITE_SIZE counters against these: the ratio counter / algorithmic is how the per-kernel FETCH multiplier is chosen (the gfx950
correction of MI355X_MICROARCH.md: wide coalesced streams are tallied at half their bytes) and how wasted traffic shows.

Workloads (P problems per GPU, S samples, N times, M inducing points, L joints, Q spheres, B bases):"""

WORKLOADS = {
    "config2": dict(P=1, S=128, N=100, M=30, L=7, Q=37, B=1024),
    "config3": dict(P=55, S=7, N=70, M=24, L=7, Q=37, B=1024),
    "config5": dict(P=64, S=128, N=100, M=30, L=14, Q=45, B=1024),
    "franka64": dict(P=64, S=128, N=100, M=30, L=7, Q=37, B=1024),
    "config4_1rank": dict(P=1, S=1024, N=70, M=18, L=6, Q=17, B=1024),
}


def workload_of(tag: str):
    """config5_mask / config5_summary / ... -> config5."""
    for name in sorted(WORKLOADS, key=len, reverse=True):
        if tag.startswith(name):
            return WORKLOADS[name]
    return None


def table(w):
    """kernel-name prefix -> dict(read=bytes, write=bytes, access=..., flops=..., note=...).  access: "wide" = 16-byte-per-lane
    coalesced streams / LDS-DMA rows (FETCH_SIZE tallies half), "gather" = scattered 16-byte records (tallied ~1:1, 65 B per
    gather that misses: profiles/r02/final/gather_probe_calib.txt), "mixed"."""
    P, S, N, M, L, Q, B = (w[k] for k in ("P", "S", "N", "M", "L", "Q", "B"))
    Mz, J, NC, PL = M + 2, N + M + 2, -(-S // 8), P * L
    # sets of partial sums the reverse pass leaves per latent: one per 8-sample chunk, or -- the register-resident kernel of large batches
    # (Mz = 32), round 5 -- one per WORKGROUP of cpw chunks (gp_path.hip: cpw doubles while that leaves kPbMinWgs workgroups)
    cpw = 1
    while cpw < NC and PL * -(-NC // (2 * cpw)) >= 1536:
        cpw *= 2
    NCp = -(-NC // cpw) if (Mz == 32 and cpw >= 2) else NC
    f4, f8 = 4, 8
    t = {}
    t["prior_fused_split_kernel"] = dict(
        read=PL * (B * L + B) * f4, write=2 * P * S * L * J * f4, access="wide", flops=2 * 2.0 * P * S * J * L * B,
        note="reads omega, beta; writes F0, H (W and the features never exist in memory)")
    t["prior_fused_small16_kernel"] = dict(
        read=PL * (B * L + B) * f4 + P * S * L * B * f4, write=4 * 2 * P * S * L * J * f4, access="wide", flops=2 * 2.0 * P * S * J * L * B,
        note="reads omega, beta, W; writes four K-slices of F0, H")
    t["paths_fwd_regs"] = dict(
        read=P * S * L * J * f4 + PL * (N * Mz + Mz * Mz + Mz) * f4 + 2 * PL * S * Mz * f4, write=P * S * L * (N + Mz) * f4, access="wide",
        flops=2.0 * P * S * L * (Mz * Mz + N * Mz), note="reads F0, AT, C, m, epsT, eps2T; writes f, R")
    t["paths_bwd_regs"] = dict(
        read=P * S * L * (N + 2 * Mz + 2 * J) * f4 + PL * N * Mz * 16 + 2 * PL * Mz * Mz * f4, write=PL * NCp * (Mz + Mz * Mz + 8) * f4,
        access="wide", flops=2.0 * P * S * L * (3 * N * Mz + 2 * Mz * Mz + Mz * Mz),
        note="reads G, R, epsT, F0, H once and A4 (+ the tangents of C) once per latent; the workgroups of a latent each fetch A4; writes ONE set of partial sums per workgroup")
    # (round 5: the rows of A -- A4, AT: N Mz 20 B per latent, 3 N Mz^2 of the float64 products -- are formed by stage A's workgroups,
    #  mid_stage1 / mid_cov_a_rng below; stage B keeps KL, the two tangents and q_sqrt)
    rows_write, rows_flops = PL * N * Mz * 20, PL * 2.0 * 4 * N * 32 * 32
    t["cov_b_kernel"] = dict(
        read=PL * (4 * Mz * Mz + M * M + M + 2 * Mz) * f8, write=PL * (5 * Mz * Mz * f4 + Mz * f4 + (M * M + M + 3) * f8),
        access="mixed", flops=PL * 2.0 * 9 * 32 ** 3,
        note="reads Kuu, dKuu, Lk, Lk^-1, q_sqrt, q_mu; writes C, CT, the two tangents, Lk32, m, the KL gradients; float64 MFMA products "
             "(nine 32^3 ones per latent: four per tangent, one for q_sqrt)")
    t["mid_hyper_final_kernel"] = dict(
        read=PL * NCp * (Mz + Mz * Mz + 8) * f4 + PL * (Mz * Mz * f4 + 4 * (M * M + M) * f8), write=PL * 4 * (M * M + M) * f8, access="wide",
        flops=PL * (2.0 * M * M * Mz / 2), note="reads the chunk partials, Lk32, KL gradients, variables + moments; writes gradient, variables, moments")
    t["mid_stage1_kernel"] = dict(t["mid_hyper_final_kernel"], note="gradient assembly of the previous step (as mid_hyper_final_kernel) beside stage A and the draws; "
                                  "+ omega / beta / eps / eps' writes + stage A's (Kuu, dKuu, Lk, Lk^-1, the inverse; A4, AT)",
                                  write=t["mid_hyper_final_kernel"]["write"] + PL * (B * L + B) * f4 + 4 * P * S * Mz * L * f4 + PL * 5 * Mz * Mz * f8 + rows_write,
                                  flops=t["mid_hyper_final_kernel"]["flops"] + PL * 2.0 * 2 * 32 ** 3 + rows_flops)
    t["mid_cov_a_rng_kernel"] = dict(read=PL * (2 * Mz) * f8, write=PL * (B * L + B) * f4 + 4 * P * S * Mz * L * f4 + PL * 5 * Mz * Mz * f8 + rows_write, access="wide",
                                     flops=PL * 2.0 * 2 * 32 ** 3 + rows_flops,
                                     note="writes omega, beta, eps / eps' in both layouts, Kuu, dKuu, Lk, Lk^-1, (Kuu + jI)^-1, A4 (16 B per entry, a quarter of it padding), AT")
    # merged launches of the batch schedule (round 5): stage B behind the prior tiles, with the path assembly as the tiles' epilogue
    # (F0 is then never read back: the epilogue works on the accumulators; what it reads instead is AT, Lk64, q_sqrt, q_mu, the eps rows)
    def merged(*parts, extra_read=0, extra_write=0, extra_flops=0.0, note=""):
        return dict(read=sum(x["read"] for x in parts) + extra_read, write=sum(x["write"] for x in parts) + extra_write, access="wide",
                    flops=sum(x["flops"] for x in parts) + extra_flops, note=note)
    fwd_read = PL * (N * Mz * f4 + Mz * Mz * f8 + (M * M + M) * f8) + 2 * PL * S * Mz * f4
    t["prior_split_cov_b_kernel"] = merged(
        t["prior_fused_split_kernel"], t["cov_b_kernel"], extra_read=fwd_read, extra_write=P * S * L * (N + Mz) * f4,
        extra_flops=2.0 * P * S * L * (Mz * Mz + N * Mz),
        note="prior tiles (reads omega, beta; writes F0, H) + their epilogue = the path assembly (reads AT, Lk, q_sqrt, q_mu, epsT, eps2T; "
             "writes f, R) + stage B's roles behind them (as cov_b_kernel)")
    t["mid_cov_b_prior16_kernel"] = merged(t["prior_fused_small16_kernel"], t["cov_b_kernel"],
                                           note="stage B's roles + the few-sample prior tiles (as prior_fused_small16_kernel) in one launch")
    lik = dict(read=P * S * L * N * f4 + 16 * P * S * N * Q, write=P * S * L * N * f4 + P * S * N * f4, access="gather",
               flops=P * S * N * (Q * 40.0 + L * 120.0),
               note="reads f and ONE 16-byte record per sphere query that is not skipped (upper bound: every query); writes G, logp; "
                    "SURVEY 8(d) prices 28 B per query: S N (28 Q + 8 L + 4) B per problem")
    for name in ("loglik_paths_mask_kernel", "loglik_paths_kernel", "loglik_paths_wide_kernel"):
        t[name] = dict(lik)
    return t


def lookup(name: str, w):
    if w is None:
        return None
    for prefix, rec in table(w).items():
        if name.startswith(prefix):
            return rec
    return None


if __name__ == "__main__":
    import json
    import sys
    w = workload_of(sys.argv[1] if len(sys.argv) > 1 else "config5")
    print(json.dumps({k: {a: (round(b / 1e6, 2) if a in ("read", "write") else b) for a, b in v.items()} for k, v in table(w).items()}, indent=1))
