// CKKS bootstrapping on the GPU evaluator: ModRaise -> (SubSum) -> CoeffsToSlots -> approximate modular
// reduction (Chebyshev cosine + double-angle) -> SlotsToCoeffs.  Mirrors what the reference obtains from
// OpenFHE's EvalBootstrapSetup / EvalBootstrapKeyGen / EvalBootstrap (reference src/FHEController.cpp
// :237-240, :280, :438-469; level budget {3,3}, 16384 slots, sparse ternary secret).
#pragma once
#include <complex>
#include <map>
#include <vector>
#include "client.h"
#include "evaluator.h"

namespace fhelin {

typedef std::complex<double> cplx;
typedef std::map<int, std::vector<cplx>> DiagMap;  // rotation (mod slots) -> diagonal: out = sum_r d_r * rot(in, r)

// one level of the homomorphic DFT, prepared for baby-step / giant-step evaluation
struct LinStage {
    int g0 = 1;    // all rotation indices are multiples of g0
    int bsz = 1;   // baby-step count
    struct Term {
        int giant;   // giant shift (slots), applied after the inner sum
        int baby;    // baby shift (slots)
        PtPtr diag;  // diagonal, pre-rotated by -giant
    };
    std::vector<Term> terms;
};

class Bootstrapper {
public:
    Bootstrapper(Evaluator& ev, Client& cl) : ev_(ev), cl_(cl) {}
    ~Bootstrapper();
    void setup(int budget_enc, int budget_dec, int slots);
    bool ready() const { return slots_ > 0; }
    // drop: raise to L+1-drop limbs only, so that the result has `drop` limbs fewer (level plan: the caller knows that the
    // circuit up to the next bootstrap leaves that many unused)
    CtPtr bootstrap(const CtPtr& ct, int drop = 0);
    // several independent ciphertexts at once (the GELU containers of one sample, src/main.cpp:354-358; the two halves of
    // affine-1, :319-320): every launch of the pipeline carries all of them, the switching keys and plaintext diagonals are
    // read once per batch.  out[i] holds exactly the residues of bootstrap(cts[i], drop).
    std::vector<CtPtr> bootstrap_batch(const std::vector<CtPtr>& cts, int drop = 0);
    // debug / test hook: stop after stage 1 (ModRaise+SubSum), 2 (CoeffsToSlots, real part), 3 (EvalMod, real part)
    CtPtr partial(const CtPtr& ct, int stage);
    int depth() const { return depth_; }
    int out_ell() const { return ev_.ctx().L + 1 - depth_; }   // limbs of the result when nothing is dropped
    // read-only views for the residue-level parity tests (include/fhelin.h fhelin_bootstrap_describe / _diag / _cheb): the
    // oracle composes the same stages from the diagonals' exported encodings
    const std::vector<LinStage>& stages(bool s2c) const { return s2c ? s2c_ : c2s_; }
    const std::vector<double>& cheb() const { return cheb_; }
    bool packed() const { return packed_; }
    int slots() const { return slots_; }

    // parameters (DESIGN.md "Bootstrapping")
    int K = 28;            // bound on |I|: t = Delta m + q0 I
    int R = 3;             // double-angle iterations
    int cheb_degree = 47;  // degree of the cosine fit (depth 6)
    int correction = 10;   // message is scaled to q0 / 2^correction before ModRaise

private:
    Evaluator& ev_;
    Client& cl_;
    int slots_ = 0;
    bool stage_order_legacy_ = false;   // FHELIN_BOOT_STAGES_LEGACY=1: larger stages first (round-1 split 5+5+4)
    bool packed_ = false;   // sparse packing: real and imaginary halves share one ciphertext through EvalMod (bootstrap.cpp)
    int depth_ = 0;
    std::vector<LinStage> c2s_, s2c_;
    std::vector<double> cheb_;
    u64* mono_i_ = nullptr;  // NTT of X^{N/2} over the Q limbs: multiplication by i

    LinStage prepare(const DiagMap& m, int n);   // n: slots the stage's diagonals are written over
    CtPtr apply(const LinStage& st, const CtPtr& x);
    CtPtr mult_i(const CtPtr& x);
    CtPtr mod_raise(const CtPtr& ct, long double& rho, int top_ell);
    std::vector<CtPtr> eval_mod(const std::vector<CtPtr>& xs);
    CtPtr run(const CtPtr& ct, int stop_after, int drop = 0);
    std::vector<CtPtr> apply_batch(const LinStage& st, const std::vector<CtPtr>& xs);
    std::vector<CtPtr> run_batch(const std::vector<CtPtr>& cts, int drop);
};

}  // namespace fhelin
