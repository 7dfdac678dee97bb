// Decoder for gnark v0.14 serialised constraint systems (`groth16.NewCS(BN254).ReadFrom`,
// gnark/libraries/prover/impl/library.go:120-125).  Format: SURVEY.md App. A.
#pragma once
#include <stdint.h>
#include <array>
#include <map>
#include <string>
#include <vector>

namespace zk {

enum BlueprintKind : int { BP_HINT = 0, BP_R1C = 1, BP_LOOKUP = 2 };

static constexpr uint32_t HINT_NBITS = 4115454955u;      // std/math/bits.nBits
static constexpr uint32_t HINT_COUNT = 2138922168u;      // std/internal/logderivarg.countHint
static constexpr uint32_t HINT_RANDOMIZE = 1774611027u;  // internal/hints.Randomize
static constexpr uint32_t HINT_BSB22 = 4156202267u;      // frontend/cs.Bsb22CommitmentComputePlaceholder
static constexpr uint32_t WIRE_CONST = 0xFFFFFFFFu;

struct R1csCommitment {
    uint32_t commitment_wire = 0;
    std::vector<uint32_t> private_committed;
    uint32_t nb_public_committed = 0;
    std::vector<uint32_t> public_and_commitment_committed;
};

struct R1cs {
    uint64_t nb_constraints = 0, nb_internal = 0;
    size_t nb_public = 0, nb_secret = 0;     // nb_public includes the constant-one wire
    std::vector<std::vector<uint32_t>> levels;
    std::vector<uint32_t> blueprint_id, constraint_offset, wire_offset;
    std::vector<uint64_t> start_calldata;
    std::vector<uint32_t> calldata;
    std::vector<std::array<uint32_t, 8>> coeffs;   // Montgomery form, as stored
    std::vector<int> bp_kind;
    std::vector<std::vector<uint32_t>> lookup_entries;   // per blueprint: EntriesCalldata
    std::vector<R1csCommitment> commitments;
    std::map<uint32_t, std::string> hint_names;

    size_t nb_wires() const { return nb_public + nb_secret + nb_internal; }
    size_t nb_instructions() const { return blueprint_id.size(); }
    int log_domain() const {
        int l = 0;
        while (((uint64_t)1 << l) < nb_constraints) l++;
        return l;
    }
};

// throws zk::ZkError on malformed input
void r1cs_parse(const uint8_t *data, size_t len, R1cs &out);

}  // namespace zk
