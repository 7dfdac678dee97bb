/* Feeds records from stdin to the C ABI of libzkprove_asan.so (host-side units built with AddressSanitizer + UBSan; see the `asan`
 * target of zk-symmetric-crypto_amd/Makefile).  Record: kind u8 | length u32 little-endian | payload.  After every record one result
 * byte goes to the file descriptor named by $FUZZ_ACK_FD when the parent passed one (stdout carries the library's own messages, which mirror the reference's
 * prints).  Any sanitizer report aborts the process (tests/test_sanitizers.py checks exit status and stderr); buffers the
 * library returns are released the way the reference's callers do (Free / VFree), so LeakSanitizer sees what is not. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <fcntl.h>
#include "zkprove.h"

static int read_exact(void *buf, size_t n) { return fread(buf, 1, n, stdin) == n; }

int main(void) {
    ZkpAllowUnpinnedKeys(1);
    const char *ack_env = getenv("FUZZ_ACK_FD");
    const int ack_fd = ack_env ? atoi(ack_env) : -1;
    const int ack = ack_fd >= 0 && fcntl(ack_fd, F_GETFD) != -1;
    uint32_t *inputs = malloc(4096 * 4);
    uint8_t *output = malloc(256);
    for (;;) {
        uint8_t kind;
        uint32_t len;
        if (!read_exact(&kind, 1) || !read_exact(&len, 4)) break;
        uint8_t *buf = malloc(len ? len : 1);              /* exact size: a read past the end is a heap overflow the sanitizer sees */
        if (len && !read_exact(buf, len)) return 2;
        GoSlice s = {buf, (long long)len, (long long)len};
        uint8_t res = 0;
        ProveReturn r;
        switch (kind) {
            case 1: {                                       /* request -> witness: JSON reader, base64, request layer, ciphers */
                long ni = 0, no = 0;
                res = (uint8_t)(zkp_request_witness(s, inputs, 4096, output, 256, &ni, &no) & 0xff);
                break;
            }
            case 2: r = Prove(s); res = r.r1 > 0; Free(r.r0); break;
            case 3: r = ProveBatch(s); res = r.r1 > 0; Free(r.r0); break;
            case 4: res = Verify(s); break;
            case 5: r = VerifyBatch(s); res = r.r1 > 0; VFree(r.r0); break;
            case 6:                                         /* payload[0] = algorithm id, rest = verifying key bytes */
                if (len >= 1) { GoSlice v = {buf + 1, (long long)len - 1, (long long)len - 1}; res = ZkpInitVerifier(buf[0], v); }
                break;
            case 7: { uint64_t st[8]; res = zkp_r1cs_stats(s, st) == 0; break; }
            case 8:                                         /* payload[0] = algorithm id, rest = gnark binary witness */
                if (len >= 1) { GoSlice w = {buf + 1, (long long)len - 1, (long long)len - 1}; r = ZkpProveWitness(buf[0], w); res = r.r1 >= 0; Free(r.r0); }
                break;
            case 9: {                                       /* payload[0] = id, u32 pk length, pk, r1cs: hash pins and header checks (no GPU here) */
                if (len >= 5) {
                    uint32_t pl;
                    memcpy(&pl, buf + 1, 4);
                    if (pl <= len - 5) {
                        GoSlice pk = {buf + 5, (long long)pl, (long long)pl}, cs = {buf + 5 + pl, (long long)(len - 5 - pl), (long long)(len - 5 - pl)};
                        res = InitAlgorithm(buf[0], pk, cs);
                    }
                }
                break;
            }
            case 10: { uint8_t out[256]; res = len >= 384 ? (uint8_t)(zkp_proof_serialize(buf, buf[0] & 1, out, sizeof(out)) & 0xff) : 0; break; }
            default: break;
        }
        free(buf);
        if (ack && write(ack_fd, &res, 1) != 1) return 3;
    }
    free(inputs);
    free(output);
    ZkpReset();
    return 0;
}
