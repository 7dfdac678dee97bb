// Batched R1CS witness solver for gfx950: one lane = one proof instance, one launch per dependency
// level of the gnark instruction graph.
//
// Replaces gnark v0.14 constraint/bn254 (*system).Solve -> solver.run / solveR1C / solveWithHint and
// the blueprints (generic R1C, generic hint, logderivlookup) -- SURVEY.md section 8 a6 -- executed inside
// groth16.Prove (gnark/libraries/prover/impl/provers.go:179,269).
#pragma once
#include "kernels.h"
#include "r1cs_reader.h"

namespace zk {

enum SolverOp : uint32_t { OP_R1C = 1, OP_NBITS = 2, OP_COUNT = 3, OP_RANDOMIZE = 4, OP_BSB22 = 5, OP_LOOKUP = 6 };

// error codes reported per instance (first error wins)
enum SolverErr : uint32_t { SERR_NONE = 0, SERR_UNSATISFIED = 1, SERR_LOOKUP_RANGE = 2, SERR_COUNT_QUERY = 3,
                            SERR_RETRY = 0xFFFFFFFFu };      // k_solve_wide_run gave up at its grid barrier: the host repeats the call with per-level launches

struct SolverLevel {
    uint32_t first = 0, count = 0;   // range in level_instr
    uint32_t wide = 0;               // the first `wide` of them need a whole workgroup (k_solve_level); the rest are packed four to one
    uint32_t n_count = 0;            // the first n_count of the wide ones are countHints (k_solve_counts)
    bool has_commit = false;         // a BSB22 hint sits in this level: commitment must be ready before launch
    bool narrow = false;             // at most 16 instructions, no countHint, no commitment: runs of such levels are one launch (k_solve_narrow_run)
    uint32_t div_first = 0, div_count = 0;   // deferred divisions of this level (records in div_list)
};

struct SolverProgram {
    size_t n_wires = 0, n_inputs = 0, n_constraints = 0, n_public = 0;
    bool constraints_covered = false;      // every constraint row is written by an R1C instruction (solver_begin clears only the padding then)
    int log_domain = 0;
    std::vector<SolverLevel> levels;
    DevBuf<uint32_t> prog, instr_off, level_instr;
    DevBuf<uint32_t> level_desc;                    // (first, count, wide, div_first, div_count) per level: what k_solve_narrow_run reads on the device
    DevBuf<Fr> coeffs, coeff_inv, lookup_tables, count_tables;   // lookup_tables: [table][256]
    DevBuf<uint32_t> div_list;                      // (constraint, loc, wire, coeff) x deferred divisions, grouped by level
    // "chain runs" (k_solve_chains): a run of narrow levels whose instructions are all multiplication rows (unknown in O, or none) and fall
    // into components that never read each other's wires -- the AES programs end in five MiMC-like permutations, 321 levels of five
    // dependent multiplications.  Every component is walked by ONE wavefront from the run's first level to its last, without a barrier:
    // chain_recs = 16 words per instruction of wave w (level order), records chain_wave_first[first + w] .. [first + w + 1];
    // chain_consts = the per-side sums of an instruction's constant terms (entry 0 = zero)
    struct ChainRun { uint32_t l0 = 0, l1 = 0, waves = 0, first = 0; };
    std::vector<ChainRun> chain_runs;
    DevBuf<uint32_t> chain_recs, chain_wave_first;
    DevBuf<Fr> chain_consts;
    // "lane program" of the small-batch kernel (k_solve_lanes; see there): per level a list of wavefront tasks -- 64 instructions of
    // ONE shape with their wire / constraint numbers as structure-of-arrays columns, or one long instruction for the lanes to share
    DevBuf<uint32_t> lp_shapes, lp_tasks, lp_soa, lp_level;
    size_t n_count_rows = 0, n_count_slots = 0;     // countHint scratch sizing: [slot][row][B]
    bool has_commit = false;
    uint32_t commit_wire = 0;
    std::vector<uint32_t> private_committed;

    void build(const R1cs &cs, hipStream_t s);
};

struct SolverBuffers {
    size_t B = 0;
    DevBuf<Fr> wires;          // [n_wires][B]
    DevBuf<Fr> a, b, c;        // [N][B]
    // The solver writes the output column c (and reads it back for the deferred divisions) but nothing after the solve stage needs it
    // on the comb path (everything done to c is folded into the key, keyfold.hip).  The further working sets of the pipelined entries
    // therefore keep only a small c of their own (the products buffer of batches <= 64) and solve into the first set's rows:
    // c_shared != nullptr replaces c.p for the solver (solve stages are serialised on one stream).
    Fr *c_shared = nullptr;
    DevBuf<uint32_t> inputs;   // [B][n_inputs] small integers (instance-major, as the host builds them)
    // the request path's compact form (request.h, RawWitness): [B][SOLVER_RAW_STRIDE] bytes -- the request's own bytes and the cipher
    // output, 320 B per proof instead of 10 KB of bit-expanded u32 (ChaCha20) -- expanded into the input wires on the device.
    // raw_alg >= 0: the next solver_begin reads `raw` (algorithm id 0 chacha20 / 1 aes-128 / 2 aes-256), else `inputs`.
    DevBuf<uint8_t> raw;
    int raw_alg = -1;
    DevBuf<Fr> randomize;      // [B] Montgomery
    DevBuf<Fr> challenge;      // [B] Montgomery (commitment challenge)
    DevBuf<uint32_t> err;      // [B] (code | constraint<<8)
    DevBuf<uint32_t> count_scratch;
    DevBuf<uint32_t> count_ticket;      // [slot][instance tile]: k_solve_counts' finished workgroups (cleared by solver_begin)
    DevBuf<uint32_t> sync;     // [0] arrival counter, [1] abort word of k_solve_wide_run's grid barrier (cleared by solver_begin)
    uint32_t sync_arrivals = 0;      // what the counter holds after the wide-run launches enqueued since solver_begin
    bool allow_wide_run = false;     // set by the one entry that can repeat a call after SERR_RETRY (Algorithm::resident_prove); the pipelined entries keep per-level launches
    void reserve(const SolverProgram &p, size_t B, size_t c_B = (size_t)-1);      // c_B: instances the own c rows are sized for (default B)
};


constexpr size_t SOLVER_RAW_STRIDE = 320;      // == request.h RAW_STRIDE
// inputs -> wires rows 0..n_inputs ; zero a,b,c ; clear errors
void solver_begin(const SolverProgram &p, SolverBuffers &buf, hipStream_t s);
// run levels [first_level, last_level)
void solver_run_levels(const SolverProgram &p, SolverBuffers &buf, size_t first_level, size_t last_level, hipStream_t s);

}  // namespace zk
