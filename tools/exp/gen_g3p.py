"""Generates tools/exp/g3p_kernel.inc: a copy of the product kernel mlp_gemm3p_kernel (dg-mesh_amd/csrc/mlp_f16x3.hpp) with
ablation / timing hooks for tools/g3_micro.hip.  VAR bit0: no MFMA, bit1: no stores, bit3: no loads, bit4: s_memtime per
phase of a tile step (0: K steps 0-7 = previous tile's epilogue, 1: step 8 = memory wait + loads + row maxima, 2: steps 9-15 =
scales + split, 3: unscale + barrier).  Investigation scratch, not product code:  python tools/exp/gen_g3p.py"""
import os

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(root, "dg-mesh_amd/csrc/mlp_f16x3.hpp")).read()
k0 = src.index("template <int EPI, bool CMAX_IN>\n__global__ void __launch_bounds__(512)\nmlp_gemm3p_kernel(")
k1 = src.index("// ---- weight gradient of the K = 256 layers")
kern = src[k0:k1]
PAD = 119


def line(text):
    return text + " " * max(1, PAD - len(text)) + "\\\n"


def rep(text, a, b, n=1):
    assert text.count(a) >= 1, a[:80]
    return text.replace(a, b)


kern = rep(kern, "template <int EPI, bool CMAX_IN>\n__global__", "template <int EPI, bool CMAX_IN, int VAR>\n__global__")
kern = ("#define ACC2_ZERO acc2 = zero16;\n" if "acc2" in kern else "#define ACC2_ZERO\n") + kern
kern = rep(kern, "mlp_gemm3p_kernel(", "g3p_kernel(")
decl = "    f32x16 acc, acc2, out;\n" if "f32x16 acc, acc2, out;" in kern else "    f32x16 acc, out;\n"
kern = rep(kern, decl,
           decl + "    unsigned long long tm_[4] = {0, 0, 0, 0}, t0_ = 0;\n"
           "#define TT(i_) if (VAR & 16) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t1_ = "
           "__builtin_amdgcn_s_memtime(); tm_[i_] += t1_ - t0_; t0_ = t1_; __builtin_amdgcn_sched_barrier(0); }\n")
# loads
kern = rep(kern, line("            R[slot_][r_] = *reinterpret_cast<const float4*>(A + (size_t)grow_ * lda + lane * 4);"),
           line("            if (!(VAR & 8)) R[slot_][r_] = *reinterpret_cast<const float4*>(A + (size_t)grow_ * lda + lane * 4);")
           + line("            else R[slot_][r_] = make_float4(1.f, 2.f, 3.f, 4.f);"))
# stores
kern = rep(kern, line("        cb[ro_ * 256] = v_;"), line("        if (!(VAR & 2) || v_ == 1234.5f) cb[ro_ * 256] = v_;"))
# MFMA
mf = [l for l in kern.split("\n") if "__builtin_amdgcn_mfma_f32_32x32x16_f16" in l]
assert len(mf) in (3, 4), len(mf)
lines_ = kern.split("\n")
i0 = lines_.index(mf[0])
i1 = lines_.index(mf[-1])
block = "\n".join(lines_[i0:i1 + 1]) + "\n"
kern = rep(kern, block, line("            if (!(VAR & 1)) {") + block +
           line('            } else { if (ks == 0) { acc = zero16; ACC2_ZERO } asm volatile("" ::"v"(fh_[ks % (PD_ + 1)]), "v"(fl_[ks % (PD_ + 1)])); }'))
# stamps
kern = rep(kern, line("            if (ks + PD_ < KS) {"), line("            if (ks == 8) TT(0)") + line("            if (ks + PD_ < KS) {"))
kern = rep(kern, line("            if (ks == 9) wave_max4_stage_a(m_[0], m_[1], m_[2], m_[3]);"),
           line("            if (ks == 8) TT(1)") + line("            if (ks == 9) wave_max4_stage_a(m_[0], m_[1], m_[2], m_[3]);"))
kern = rep(kern, line("        /* unscale into `out` (stored during the next step) */"),
           line("        TT(2)") + line("        /* unscale into `out` (stored during the next step) */"))
kern = rep(kern, line('        asm volatile("s_waitcnt lgkmcnt(0)\\n\\ts_barrier" ::: "memory");') + "    }\n",
           line('        asm volatile("s_waitcnt lgkmcnt(0)\\n\\ts_barrier" ::: "memory");') + line("        TT(3)") + "    }\n")
kern = rep(kern, "    P3_STEP(0, 1, true)\n", "    if (VAR & 16) t0_ = __builtin_amdgcn_s_memtime();\n    P3_STEP(0, 1, true)\n")
kern = rep(kern, "    if (colmax != nullptr) {\n        const float o = __shfl_xor(cmax, 32, 64);",
           "    if ((VAR & 16) && blockIdx.x == 0 && (wv == 0 || wv == 4) && lane == 0) {\n"
           "        unsigned long long* dbg_ = reinterpret_cast<unsigned long long*>(colmax + 256);\n"
           "        for (int i = 0; i < 4; i++) dbg_[(wv >> 2) * 8 + i] = tm_[i];\n"
           "        dbg_[(wv >> 2) * 8 + 5] = (unsigned long long)my_tiles;\n    }\n"
           "    if (colmax != nullptr) {\n        const float o = __shfl_xor(cmax, 32, 64);")
# bit5 (32): no epilogue / maxima / split work (MFMA + fragment reads + barrier only); bit6 (64): no fragment reads
kern = rep(kern, line("            if (!(FIRST_) && ks < 8) {"), line("            if (!(FIRST_) && ks < 8 && !(VAR & 32)) {"))
kern = rep(kern, line("            if (ks == 8) {") + line("                P3_ROWMAX(SN_) /* first use of tile j+1's rows: the tile's one memory wait */"),
           line("            if (ks == 8 && !(VAR & 32)) {") + line("                P3_ROWMAX(SN_) /* first use of tile j+1's rows: the tile's one memory wait */"))
kern = rep(kern, line("            if (ks == 9) wave_max4_stage_a(m_[0], m_[1], m_[2], m_[3]);"), line("            if (ks == 9 && !(VAR & 32)) wave_max4_stage_a(m_[0], m_[1], m_[2], m_[3]);"))
kern = rep(kern, line("            if (ks == 10) wave_max4_stage_b(m_[0], m_[1], m_[2], m_[3]);"), line("            if (ks == 10 && !(VAR & 32)) wave_max4_stage_b(m_[0], m_[1], m_[2], m_[3]);"))
kern = rep(kern, line("            if (ks == 11) P3_SCALES(pb ^ 1)"), line("            if (ks == 11 && !(VAR & 32)) P3_SCALES(pb ^ 1)"))
kern = rep(kern, line("            if (ks >= 12) P3_SPLIT_ROW(SN_, pb ^ 1, ks - 12)"), line("            if (ks >= 12 && !(VAR & 32)) P3_SPLIT_ROW(SN_, pb ^ 1, ks - 12)"))
kern = kern.replace("as_f16x8(*reinterpret_cast<const uint4*>(ps_", "as_f16x8((VAR & 64) ? make_uint4(0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u) : *reinterpret_cast<const uint4*>(ps_")
kern = kern.replace("#undef P3_STEP\n", "#undef P3_STEP\n#undef TT\n")
open(os.path.join(root, "tools/exp/g3p_kernel.inc"), "w").write(kern)
print("wrote tools/exp/g3p_kernel.inc", len(kern))
