"""Generates tools/exp/g3_kernel.inc: a copy of the product kernel mlp_gemm3r_kernel (dg-mesh_amd/csrc/mlp_f16x3.hpp)
with ablation / timing hooks for tools/g3_micro.hip.  VAR bit0: no MFMA, bit1: no stores, bit3: no loads, bit4: per-phase
s_memtime.  Investigation scratch, not product code:  python tools/exp/gen_g3.py"""
import os

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(root, "dg-mesh_amd/csrc/mlp_f16x3.hpp")).read()
k0 = src.index("template <int EPI, int KS, int PF, bool DUAL = false>\n__global__")
k1 = src.index("// ---- the K = 256 layer GEMM, software-pipelined")
kern = src[k0:k1]
kern = kern.replace("template <int EPI, int KS, int PF, bool DUAL = false>\n__global__", "template <int EPI, int KS, int PF, int VAR, bool DUAL = false>\n__global__")
kern = kern.replace("mlp_gemm3r_kernel(", "g3_kernel(")
PAD = 119


def line(text):
    return text + " " * max(1, PAD - len(text)) + "\\\n"


def rep(text, a, b):
    assert a in text, a[:70]
    return text.replace(a, b)


mf = [l for l in kern.split("\n") if "__builtin_amdgcn_mfma_f32_32x32x16_f16" in l]
assert len(mf) == 12  # six of the DUAL branch, six of the two-chain branch
lines_ = kern.split("\n")
i0_ = lines_.index(mf[0]) - 1
i1_ = lines_.index(mf[-1]) + 1
block = "\n".join(lines_[i0_:i1_ + 1]) + "\n"
kern = rep(kern, block, line("            if (!(VAR & 1)) {") + block +
           line('            } else { if (ks == 0) { acc = zero16; acc2 = zero16; } asm volatile("" ::"v"(fh_[ks % 3]), "v"(fl_[ks % 3])); }'))
kern = kern.replace("gemm3r_store<EPI, true>(acc,", "g3_store<EPI, true, VAR>(acc,").replace("gemm3r_store<EPI, false>(acc,", "g3_store<EPI, false, VAR>(acc,")
ld = [l for l in kern.split("\n") if "if (K % 256 == 0 || k_ < K) {" in l]
kern = rep(kern, ld[0] + "\n", line("                if ((K % 256 == 0 || k_ < K) && !(VAR & 8)) {"))
kern = rep(kern, line("        wave_max4_nonneg_lane63(m_[0], m_[1], m_[2], m_[3]);"),
           line("        if (!(VAR & 32)) wave_max4_nonneg_lane63(m_[0], m_[1], m_[2], m_[3]);"))
kern = rep(kern, line("                    *reinterpret_cast<uint2*>(d_ + 2 * k_) = make_uint2(h0_, h1_);"),
           line("                    if (!(VAR & 64) || h0_ == 0x12345u) *reinterpret_cast<uint2*>(d_ + 2 * k_) = make_uint2(h0_, h1_);"))
kern = rep(kern, line("                    *reinterpret_cast<uint2*>(d_ + PLANE + 2 * k_) = make_uint2(l0_, l1_);"),
           line("                    if (!(VAR & 64) || l0_ == 0x12345u) *reinterpret_cast<uint2*>(d_ + PLANE + 2 * k_) = make_uint2(l0_, l1_);"))
kern = rep(kern, line("            rinv[(pb_) * 32 + row_] = inv_; /* every lane writes the same word: no exec juggling */"),
           line("            if (!(VAR & 128) || lane == 0) rinv[(pb_) * 32 + row_] = inv_;"))
# timing hooks
kern = rep(kern, "    f32x16 acc, acc2;\n    const f32x16 zero16",
           "    unsigned long long tm_[6] = {0, 0, 0, 0, 0, 0}, t0_ = 0;\n#define TT(i_) if (VAR & 16) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t1_ = "
           "__builtin_amdgcn_s_memtime(); tm_[i_] += t1_ - t0_; t0_ = t1_; __builtin_amdgcn_sched_barrier(0); }\n    f32x16 acc, acc2;\n    const f32x16 zero16")
kern = rep(kern, line("#define R3_SPLIT(slot_, pb_)") + line("    {"),
           line("#define R3_SPLIT(slot_, pb_)") + line("    {") + line('        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); TT(4)'))
lines = kern.split("\n")
out = []
in_step = False
branch = None
for l in lines:
    s = l.rstrip("\\").rstrip()
    if "#define R3_STEP(" in l:
        in_step = True
    if in_step:
        if s.strip() == "if (wv < 4) {":
            branch = "A"
        elif s.strip() == "} else {":
            branch = "B"
        if s.strip() == "R3_UNSCALE(pb)":
            out.append(l)
            out.append(line("            TT(1)").rstrip("\n"))
            continue
        if branch == "A" and s.strip() == "R3_STORE(tile)":
            out.append(line("            TT(0)").rstrip("\n"))
            out.append(l)
            out.append(line("            TT(2)").rstrip("\n"))
            continue
        if branch == "B" and s.strip() == "if ((j_) > 0) R3_STORE(tile - G)":
            out.append(line("            TT(0)").rstrip("\n"))
            out.append(l)
            out.append(line("            TT(2)").rstrip("\n"))
            continue
        if s.strip() == "R3_LDS_BARRIER();":
            out.append(l)
            out.append(line("        TT(3)").rstrip("\n"))
            continue
        if s.strip() == "}" and not l.rstrip().endswith("\\"):
            in_step = False
    out.append(l)
kern = "\n".join(out)
kern = rep(kern, "    for (int j = 0; j < my_tiles; j += 2) {\n        R3_STEP(j, PF - 1)",
           "    if (VAR & 16) t0_ = __builtin_amdgcn_s_memtime();\n    for (int j = 0; j < my_tiles; j += 2) {\n        R3_STEP(j, PF - 1)")
kern = rep(kern, "    if (colmax != nullptr) {\n        // lanes li and li + 32",
           "    if ((VAR & 16) && blockIdx.x == 0 && (wv == 0 || wv == 4) && lane == 0) {\n        unsigned long long* dbg_ = reinterpret_cast<unsigned long long*>(colmax + 256);\n"
           "        for (int i = 0; i < 5; i++) dbg_[(wv >> 2) * 8 + i] = tm_[i];\n        dbg_[(wv >> 2) * 8 + 5] = (unsigned long long)my_tiles;\n    }\n"
           "    if (colmax != nullptr) {\n        // lanes li and li + 32")
e0 = src.index("// second half: bias / ReLU / mask bits")
epi = src[e0:k0].replace("template <int EPI, bool FULL>", "template <int EPI, bool FULL, int VAR>").replace("gemm3r_store(", "g3_store(")
epi = rep(epi, "            if (FULL || row0 + ro < M) {\n                cb[ro * 256] = v;",
          "            if ((FULL || row0 + ro < M) && (!(VAR & 2) || v == 1234.5f)) {\n                cb[ro * 256] = v;")
epi = rep(epi, "        if (FULL || row0 + ro < M) mb[ro * 8] = mwsel;", "        if ((FULL || row0 + ro < M) && (!(VAR & 2) || mwsel == 0x12345u)) mb[ro * 8] = mwsel;")
open(os.path.join(root, "tools/exp/g3_kernel.inc"), "w").write(epi + kern)
print("wrote tools/exp/g3_kernel.inc")
