// Groth16 batch prover for one circuit ("algorithm"): device-resident proving key, solver program,
// NTT domain and the batch pipeline  solve -> [commit] -> computeH -> 5 (+2) MSMs -> finalize.
//
// Replaces groth16.Prove(r1cs, pk, witness) of gnark v0.14 (call sites
// gnark/libraries/prover/impl/provers.go:179,269) and ProvingKey.ReadFrom (impl/library.go:104-109).
#pragma once
#include <atomic>
#include <functional>
#include <memory>
#include <mutex>
#include "hostutil.h"
#include "solver.h"

namespace zk {

struct ProvingKeyDev {
    uint64_t domain = 0;
    size_t n_wires = 0;
    DevBuf<G1Affine> g1_misc;   // alpha, beta, delta
    DevBuf<G2Affine> g2_misc;   // beta, delta
    DevBuf<G1Affine> delta_tab; // fixed-base table of delta (8-bit windows)
    DevBuf<G2Affine> delta2_tab;
    MsmBasesG1 A, B1, K, Z, ped, ped_sigma;   // Z and K hold pk.G1.Z / pk.G1.K in the folded form of keyfold.hip
    MsmBasesG2 B2;
    size_t n_ped = 0;
    // quotient MSM of large batches: comb tables over the evaluation-form bases Ze (msm_comb.hip), built the first time a batch of
    // at least comb_min_batch() proofs is reserved (tens of GB and ~1-3 s: not spent on keys that only ever serve single Prove() calls)
    DevBuf<G1Affine> ze;        // this rank's slice of Ze (kept for the lazy build)
    CombBases Zcomb;
    bool comb_tried = false;
};

// canonical little-endian affine coordinates; (0,0) = infinity
struct ProofPoints {
    U256 ar[2], bs[4] /* x.a0 x.a1 y.a0 y.a1 */, krs[2], commitment[2], pok[2];
};

// A witness element that does not fit the u32 assignment the request layer packs (ZkpProveWitness accepts any gnark witness vector:
// 32-byte field elements): written over the solver's input row after the u32 assignment.  value: canonical.
struct WideInput { uint32_t instance, index; U256 value; };

struct PhaseTimes {   // milliseconds, device time measured with HIP events on the prover's stream
    float solve = 0, commit = 0, ntt = 0, msm_g1 = 0, msm_g2 = 0, finalize = 0, total = 0;
    float z_segments = 0;   // the dominant kernel alone: k_msm_accumulate<Fp, true> of the quotient MSM
};

// Source / sink of the device chunks of a streamed batch (Algorithm::prove_stream): the host side hands over chunks as it finishes
// preparing them and takes the proofs back chunk by chunk, so request decoding / witness packing of later chunks and the proof
// serialisation of earlier ones overlap the device work (cabi.cpp, ProveBatch).
struct ChunkFeed {
    struct Chunk {
        size_t id = 0, count = 0;                 // count <= Algorithm::max_chunk
        const uint32_t *inputs = nullptr;         // [count][n_inputs]  -- or, the request path's compact form:
        const uint8_t *raw = nullptr;             // [count][SOLVER_RAW_STRIDE] records (request.h RawWitness), expanded on the device
        int raw_alg = -1;                         // algorithm id the records are laid out for (raw != nullptr)
        const U256 *r = nullptr, *s = nullptr, *rnd = nullptr;
        ProofPoints *out = nullptr;               // [count], written before done()
        uint32_t *errs = nullptr;
    };
    virtual bool next(Chunk &c) = 0;              // blocks until the next chunk is ready; false: no more chunks
    virtual void done(const Chunk &c) = 0;        // the chunk's proofs and solver error words have been written
    virtual ~ChunkFeed() {}
};

class Algorithm {
   public:
    R1cs cs;
    SolverProgram prog;
    ProvingKeyDev pk;
    NttDomain dom;
    bool has_commit = false;
    size_t commit_level = 0;
    int device = 0;          // the HIP device that holds this replica's key and buffers; every entry below binds it on the calling thread
    // Split mode (SURVEY 8e "single-proof latency mode"): this process is rank split_rank of split_world and keeps slice split_rank
    // of every base set of the key (the Pedersen bases stay whole: the commitment is needed in the middle of the solve).  Every rank
    // solves and transforms the whole batch, multiplies its slices (split_partials), the ranks exchange the five partial sums per
    // proof (RCCL / gloo all_gather by the caller) and each finishes the proof from the gathered sums (split_finish).
    int comb_share = 1;      // replicas the library keeps on this device: the comb tables' memory budget is divided by it (cabi.cpp)
    int split_rank = 0, split_world = 1;
    size_t z_offset = 0;     // first coset point of this rank's slice of the quotient bases
    void split_partials(uint8_t *out384);                       // [B][384]: Ar, Bs1, K, Z (64 B each), Bs2 (128 B), affine Montgomery LE
    void split_finish(int world, const uint8_t *gathered);      // [world][B][384] -> proofs (then resident_download)

    // throws ZkError
    void load(const uint8_t *pk_bytes, size_t pk_len, const uint8_t *r1cs_bytes, size_t r1cs_len);

    // Batch prove.  inputs: [batch][n_inputs] small integers (witness assignment: public without the
    // constant one, then secret).  r, s, randomize: [batch] canonical Fr (randomize may be null when
    // the circuit has no Randomize hint).  Results in out[batch]; errs[batch] = solver error words
    // (0 = ok).  Work is processed in chunks of at most `chunk` instances.
    // wide (optional, single-chunk batches only): elements of the assignment that are wider than 32 bits
    void prove_batch(size_t batch, const uint32_t *inputs, const U256 *r, const U256 *s, const U256 *randomize, ProofPoints *out,
                     uint32_t *errs, PhaseTimes *times = nullptr, const WideInput *wide = nullptr, size_t n_wide = 0);
    // the same from the request path's records: raw = [batch][SOLVER_RAW_STRIDE] bytes laid out for algorithm raw_alg
    void prove_batch_raw(size_t batch, const uint8_t *raw, int raw_alg, const U256 *r, const U256 *s, const U256 *randomize, ProofPoints *out, uint32_t *errs);
    void prove_batch_impl(size_t batch, const uint32_t *inputs, const uint8_t *raw, int raw_alg, const U256 *r, const U256 *s, const U256 *randomize, ProofPoints *out,
                          uint32_t *errs, PhaseTimes *times, const WideInput *wide, size_t n_wide);

    // the same over chunks pulled from a feed (several replicas may pull from one feed: dynamic sharing of a batch between devices)
    void prove_stream(ChunkFeed &feed, PhaseTimes *times = nullptr);

    // device-resident variant for the benchmark: inputs already in HBM as [n_inputs][B] (see bench.py)
    struct Resident;
    void upload_set(struct SolveSet &S, size_t B, const uint32_t *inputs, const U256 *r, const U256 *s, const U256 *randomize, hipStream_t st,
                    const uint8_t *raw = nullptr, int raw_alg = -1);
    void ensure_second_set(int nsets = 2);
    void ensure_second_set_once(int nsets);
    void enqueue_solve(struct SolveSet &S, MsmScratch &commit_scratch, hipStream_t st);
    void enqueue_main(struct SolveSet &S, hipStream_t st, hipEvent_t before_msm, hipStream_t tail);
    bool ahead_eligible() const;
    // Wire comb of the AES circuits (msm.hip, MsmBases::wcomb): a calibration solve over `calib_inputs` (synthetic consistent witness
    // assignments supplied by the request layer, cabi.cpp) classifies every wire position by the width of its values; the positions
    // that are not 0 / +-1 get comb tables per base set.  Done once, with the first batch of at least comb_min_batch() proofs.
    std::vector<uint32_t> calib_inputs;                     // [n][n_inputs]
    std::vector<U256> calib_rnd;                            // [n] Randomize hint values
    bool wire_comb_tried = false, wire_comb_on = false;
    bool comb_dropped = false;          // the tables went under memory pressure: the next proving call looks at the budget again (comb_second_chance)
    int comb_rebuilds = 0;              // ... at most twice in an algorithm's life (a build costs seconds: no ping-pong between circuits)
    void comb_second_chance();
    void wire_comb_prepare(size_t B);
    void wire_comb_reserve(MsmScratch &sc, bool main_set, size_t B);
    void debug_leftovers(long *out3);                       // {sum, max, instances with any} of the compact bucket-pipeline list left by the last main-stream wire MSM
    size_t resident_batch() const;                          // instances of the current resident batch (0: none)
    bool comb_active() const;                               // the current resident batch takes the comb path for its quotient MSM
    void comb_prepare(size_t B);
    bool comb_drop_tables();
    void resident_reserve_buffers(size_t B);
    size_t own_c_rows(size_t B) const;
    size_t side_max_b() const;                              // batches up to this size run their MSM chains side by side (prover.hip)
    bool comb_release();                                    // out of memory: give the comb tables back, continue on the bucket pipeline                            // lazy table build + scratch (resident_reserve)
    // Memory pressure across algorithms: the hook (set by cabi.cpp, which owns the registry) asks every OTHER algorithm on this device
    // that is idle right now to give its comb tables back (release_comb_if_idle); true when something was freed
    static bool (*neighbours_release)(Algorithm *self);
    static int (*loaded_on_device)(int device);             // replicas of ALL algorithms the registry holds on a device (cabi.cpp): the comb budget's divisor
    static bool (*peers_visit)(Algorithm *self, const std::function<void(Algorithm *)> &fn);      // every other replica on self's device (cabi.cpp)
    // Filling the device (comb_prepare): once every circuit loaded on the device has reserved its working sets and built its tables, what
    // is still free goes into the tables -- a prefix of clumps one base larger (CombShape::n1) -- shared between the circuits by relative
    // gain per byte.  One rebuild per algorithm.
    int wide_run_strikes = 0;                               // grid-barrier timeouts of the one-call solver (resident_prove)
    bool comb_topped = false;
    struct CombPeer { bool settled = false, can_top = false; size_t pending = 0, weight = 0; };
    CombPeer comb_peer_view();                              // caller holds this->mu (try_lock)
    size_t pending_bytes() const;                           // what this replica will still allocate at its reserved capacity (further working sets, commitment scratch)
    void comb_top_up();                                     // proving entries: rebuild the quotient tables larger when the device has settled
    bool release_comb_if_idle();
    size_t comb_table_bytes() const;                        // quotient + wire comb tables held right now
    std::atomic<uint64_t> last_use{0};                      // stamp of the last proving call (prove_batch / prove_stream)
    bool relieve_pressure(bool rereserve);                  // own tables first (rereserve: + the bucket pipeline's buffers, comb_release), one idle neighbour's next
    void enqueue_ntt_sort(struct SolveSet &S, hipStream_t sn, int slot, hipEvent_t slot_free = nullptr);
    hipStream_t side_chain_stream(hipStream_t solve_stream, struct SolveSet &S);
    void enqueue_patterns_late(struct SolveSet &S, hipEvent_t walk_done);
    void enqueue_tail_early(struct SolveSet &S, hipStream_t st);
    void enqueue_finalize(struct SolveSet &S, hipStream_t st);
    void collect_times(struct SolveSet &S, PhaseTimes *times);
    std::unique_ptr<Resident> res;
    void resident_reserve(size_t B);
    void resident_upload(size_t B, const uint32_t *inputs_rowmajor_instance, const U256 *r, const U256 *s, const U256 *randomize, const WideInput *wide = nullptr, size_t n_wide = 0,
                         const uint8_t *raw = nullptr, int raw_alg = -1);
    void resident_prove(PhaseTimes *times);                  // one batch, one stream
    void resident_prove_pipelined(int nsteps, PhaseTimes *times);   // timed region of bench.py: solve(k+1) under NTT+MSM(k)
    void resident_download(ProofPoints *out, uint32_t *errs);
    void debug_fetch(const uint32_t *instances, size_t n, uint8_t *wires_out, uint8_t *h_out);   // parity tests (nullptr: every instance)

    size_t n_inputs() const { return prog.n_inputs; }
    size_t max_chunk = 1024;
    std::mutex mu;            // one batch at a time per algorithm (device buffers are shared)
    ~Algorithm();
    Algorithm();
};

// gnark groth16 Proof.WriteTo (compressed points): Ar | Bs | Krs | u32 nCommit | commitments | PoK
std::vector<uint8_t> proof_serialize(const ProofPoints &p, bool has_commit);

}  // namespace zk
