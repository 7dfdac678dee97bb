/* N-API binding of the libprove / libverify C ABI (include/zkprove.h) for Node.
 *
 * The reference loads bin/gnark/<platform>-<arch>-libprove.so with koffi and declares GoSlice / ProveReturn by hand
 * (js/src/gnark/utils.ts:47-109).  koffi is not in this image, so this file is the same binding as a small N-API
 * addon: it dlopen()s whichever library path the JS side hands it and calls the identical symbols with the identical
 * structs.  Nothing here knows about GPUs; a library built from the reference's Go sources would load the same way.
 *
 * Build: js/Makefile (gcc, node headers from /usr/include/node).
 */
#define NAPI_VERSION 4
#include <node_api.h>
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { void *data; long long len; long long cap; } GoSlice;
typedef struct { void *r0; long long r1; } ProveReturn;

typedef unsigned char (*fn_init_t)(unsigned char, GoSlice, GoSlice);
typedef ProveReturn (*fn_prove_t)(GoSlice);
typedef void (*fn_free_t)(void *);
typedef unsigned char (*fn_verify_t)(GoSlice);
typedef unsigned char (*fn_initvk_t)(unsigned char, GoSlice);
typedef void (*fn_allow_t)(int);
typedef int (*fn_setup_t)(GoSlice, const uint8_t *, int, void **, long long *, void **, long long *);
typedef const char *(*fn_err_t)(void);

static struct {
    void *h;
    fn_init_t init;
    fn_prove_t prove, prove_batch, verify_batch;
    fn_free_t free_, vfree;
    fn_verify_t verify;
    fn_initvk_t init_vk;
    fn_allow_t allow;
    fn_setup_t setup;
    fn_err_t last_error;
} L;

#define CHECK(env, call)                                                        \
    do {                                                                        \
        if ((call) != napi_ok) {                                                \
            napi_throw_error((env), NULL, "zkprove_napi: N-API call failed");   \
            return NULL;                                                        \
        }                                                                       \
    } while (0)

static napi_value throw_msg(napi_env env, const char *msg) {
    napi_throw_error(env, NULL, msg);
    return NULL;
}

/* Buffer / Uint8Array / string -> (pointer, length); strings are copied into *owned (caller frees) */
static int get_bytes(napi_env env, napi_value v, void **data, size_t *len, void **owned) {
    bool is_ta = false;
    *owned = NULL;
    if (napi_is_typedarray(env, v, &is_ta) == napi_ok && is_ta) {
        napi_typedarray_type t;
        napi_value ab;
        size_t off;
        if (napi_get_typedarray_info(env, v, &t, len, data, &ab, &off) != napi_ok) return -1;
        if (t != napi_uint8_array) return -1;
        return 0;
    }
    napi_valuetype vt;
    if (napi_typeof(env, v, &vt) != napi_ok || vt != napi_string) return -1;
    size_t n = 0;
    if (napi_get_value_string_utf8(env, v, NULL, 0, &n) != napi_ok) return -1;
    char *buf = (char *)malloc(n + 1);
    if (!buf) return -1;
    if (napi_get_value_string_utf8(env, v, buf, n + 1, &n) != napi_ok) { free(buf); return -1; }
    *data = buf; *len = n; *owned = buf;
    return 0;
}

static void *sym(const char *name) { return L.h ? dlsym(L.h, name) : NULL; }

/* open(path) -> { extensions: bool }   (extensions = the library also exports ProveBatch / ZkpSetup / ...) */
static napi_value js_open(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1];
    CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    char path[4096];
    size_t n = 0;
    if (argc < 1 || napi_get_value_string_utf8(env, argv[0], path, sizeof path, &n) != napi_ok) return throw_msg(env, "open(path): path expected");
    void *h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
        char msg[4400];
        snprintf(msg, sizeof msg, "Gnark library not built for OS/arch: %s", dlerror());
        return throw_msg(env, msg);
    }
    L.h = h;
    L.init = (fn_init_t)sym("InitAlgorithm");
    L.prove = (fn_prove_t)sym("Prove");
    L.free_ = (fn_free_t)sym("Free");
    L.verify = (fn_verify_t)sym("Verify");
    L.vfree = (fn_free_t)sym("VFree");
    L.prove_batch = (fn_prove_t)sym("ProveBatch");
    L.verify_batch = (fn_prove_t)sym("VerifyBatch");
    L.init_vk = (fn_initvk_t)sym("ZkpInitVerifier");
    L.allow = (fn_allow_t)sym("ZkpAllowUnpinnedKeys");
    L.setup = (fn_setup_t)sym("ZkpSetup");
    L.last_error = (fn_err_t)sym("zkp_last_error");
    if (!L.init || !L.prove || !L.free_) return throw_msg(env, "library does not export InitAlgorithm / Prove / Free");
    napi_value out, b;
    CHECK(env, napi_create_object(env, &out));
    CHECK(env, napi_get_boolean(env, L.prove_batch != NULL, &b));
    CHECK(env, napi_set_named_property(env, out, "extensions", b));
    CHECK(env, napi_get_boolean(env, L.verify != NULL, &b));
    CHECK(env, napi_set_named_property(env, out, "verifier", b));
    CHECK(env, napi_get_boolean(env, L.verify_batch != NULL, &b));
    CHECK(env, napi_set_named_property(env, out, "batchVerifier", b));
    return out;
}

/* initAlgorithm(id, pk, r1cs) -> bool */
static napi_value js_init(napi_env env, napi_callback_info info) {
    size_t argc = 3;
    napi_value argv[3];
    CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    if (!L.init) return throw_msg(env, "library not opened");
    uint32_t id;
    void *pk, *r1, *o1, *o2;
    size_t npk, nr1;
    if (argc < 3 || napi_get_value_uint32(env, argv[0], &id) != napi_ok || get_bytes(env, argv[1], &pk, &npk, &o1)) return throw_msg(env, "initAlgorithm(id, pk, r1cs)");
    if (get_bytes(env, argv[2], &r1, &nr1, &o2)) { free(o1); return throw_msg(env, "initAlgorithm(id, pk, r1cs)"); }
    GoSlice a = {pk, (long long)npk, (long long)npk}, b = {r1, (long long)nr1, (long long)nr1};
    unsigned char ok = L.init((unsigned char)id, a, b);
    free(o1); free(o2);
    napi_value out;
    CHECK(env, napi_get_boolean(env, ok != 0, &out));
    return out;
}

static napi_value call_json(napi_env env, napi_callback_info info, fn_prove_t fn, const char *what) {
    size_t argc = 1;
    napi_value argv[1];
    CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    if (!fn) return throw_msg(env, what);
    void *p, *own;
    size_t n;
    if (argc < 1 || get_bytes(env, argv[0], &p, &n, &own)) return throw_msg(env, "expected a Buffer, Uint8Array or string");
    GoSlice s = {p, (long long)n, (long long)n};
    ProveReturn r = fn(s);
    free(own);
    napi_value out;
    void *copy;
    napi_status st = napi_create_buffer_copy(env, (size_t)r.r1, r.r0, &copy, &out);
    L.free_(r.r0);      /* avoid the leak the reference warns about (utils.ts:223) */
    CHECK(env, st);
    return out;
}
/* prove(params) -> Buffer holding the response JSON */
static napi_value js_prove(napi_env env, napi_callback_info info) { return call_json(env, info, L.prove, "library not opened"); }
/* proveBatch(paramsArrayJson) -> Buffer holding a JSON array of responses */
static napi_value js_prove_batch(napi_env env, napi_callback_info info) { return call_json(env, info, L.prove_batch, "library has no ProveBatch"); }

/* ---- the same two calls off the main thread (libuv pool), returning a Promise */
typedef struct {
    napi_async_work work;
    napi_deferred deferred;
    fn_prove_t fn;
    void *in;
    size_t in_len;
    ProveReturn out;
} AsyncCall;

static void async_exec(napi_env env, void *data) {
    (void)env;
    AsyncCall *c = (AsyncCall *)data;
    GoSlice s = {c->in, (long long)c->in_len, (long long)c->in_len};
    c->out = c->fn(s);
}
static void async_done(napi_env env, napi_status status, void *data) {
    AsyncCall *c = (AsyncCall *)data;
    napi_value buf;
    void *copy;
    if (status == napi_ok && napi_create_buffer_copy(env, (size_t)c->out.r1, c->out.r0, &copy, &buf) == napi_ok) {
        napi_resolve_deferred(env, c->deferred, buf);
    } else {
        napi_value msg, err;
        napi_create_string_utf8(env, "zkprove_napi: asynchronous call failed", NAPI_AUTO_LENGTH, &msg);
        napi_create_error(env, NULL, msg, &err);
        napi_reject_deferred(env, c->deferred, err);
    }
    if (c->out.r0) L.free_(c->out.r0);
    napi_delete_async_work(env, c->work);
    free(c->in);
    free(c);
}
static napi_value call_json_async(napi_env env, napi_callback_info info, fn_prove_t fn, const char *what) {
    size_t argc = 1;
    napi_value argv[1];
    CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    if (!fn) return throw_msg(env, what);
    void *p, *own;
    size_t n;
    if (argc < 1 || get_bytes(env, argv[0], &p, &n, &own)) return throw_msg(env, "expected a Buffer, Uint8Array or string");
    AsyncCall *c = (AsyncCall *)calloc(1, sizeof *c);
    if (!c) { free(own); return throw_msg(env, "out of memory"); }
    c->fn = fn;
    c->in = malloc(n ? n : 1);
    if (!c->in) { free(own); free(c); return throw_msg(env, "out of memory"); }
    memcpy(c->in, p, n);          /* the JS buffer may be collected or mutated while the call runs */
    c->in_len = n;
    free(own);
    napi_value promise, name;
    CHECK(env, napi_create_promise(env, &c->deferred, &promise));
    CHECK(env, napi_create_string_utf8(env, "zkprove", NAPI_AUTO_LENGTH, &name));
    CHECK(env, napi_create_async_work(env, NULL, name, async_exec, async_done, c, &c->work));
    CHECK(env, napi_queue_async_work(env, c->work));
    return promise;
}
static napi_value js_prove_async(napi_env env, napi_callback_info info) { return call_json_async(env, info, L.prove, "library not opened"); }
static napi_value js_prove_batch_async(napi_env env, napi_callback_info info) { return call_json_async(env, info, L.prove_batch, "library has no ProveBatch"); }
/* verifyBatchAsync(jsonArrayOfVerifyRequests) -> Promise<Buffer> holding a JSON array of booleans (the library's result is released with
 * Free, which is the same free() as VFree) */
static napi_value js_verify_batch_async(napi_env env, napi_callback_info info) { return call_json_async(env, info, L.verify_batch, "library has no VerifyBatch"); }

/* verify(params) -> bool */
static napi_value js_verify(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1];
    CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    if (!L.verify) return throw_msg(env, "library has no Verify");
    void *p, *own;
    size_t n;
    if (argc < 1 || get_bytes(env, argv[0], &p, &n, &own)) return throw_msg(env, "expected a Buffer, Uint8Array or string");
    GoSlice s = {p, (long long)n, (long long)n};
    unsigned char ok = L.verify(s);
    free(own);
    napi_value out;
    CHECK(env, napi_get_boolean(env, ok == 1, &out));
    return out;
}

/* initVerifier(id, vk) -> bool */
static napi_value js_init_vk(napi_env env, napi_callback_info info) {
    size_t argc = 2;
    napi_value argv[2];
    CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    if (!L.init_vk) return throw_msg(env, "library has no ZkpInitVerifier");
    uint32_t id;
    void *p, *own;
    size_t n;
    if (argc < 2 || napi_get_value_uint32(env, argv[0], &id) != napi_ok || get_bytes(env, argv[1], &p, &n, &own)) return throw_msg(env, "initVerifier(id, vk)");
    GoSlice s = {p, (long long)n, (long long)n};
    unsigned char ok = L.init_vk((unsigned char)id, s);
    free(own);
    napi_value out;
    CHECK(env, napi_get_boolean(env, ok != 0, &out));
    return out;
}

/* allowUnpinnedKeys(bool) */
static napi_value js_allow(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1];
    CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    bool b = false;
    if (argc >= 1) napi_get_value_bool(env, argv[0], &b);
    if (L.allow) L.allow(b ? 1 : 0);
    return NULL;
}

/* setup(r1cs, toxic192, raw) -> { pk: Buffer, vk: Buffer }   (keygen.go:94-132 on the device) */
static napi_value js_setup(napi_env env, napi_callback_info info) {
    size_t argc = 3;
    napi_value argv[3];
    CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    if (!L.setup) return throw_msg(env, "library has no ZkpSetup");
    void *r1, *tox, *o1, *o2;
    size_t nr1, ntox;
    bool raw = true;
    if (argc < 2 || get_bytes(env, argv[0], &r1, &nr1, &o1)) return throw_msg(env, "setup(r1cs, toxic192, raw)");
    if (get_bytes(env, argv[1], &tox, &ntox, &o2) || ntox != 192) { free(o1); return throw_msg(env, "setup: toxic192 must be 192 bytes"); }
    if (argc >= 3) napi_get_value_bool(env, argv[2], &raw);
    void *pk = NULL, *vk = NULL;
    long long npk = 0, nvk = 0;
    GoSlice s = {r1, (long long)nr1, (long long)nr1};
    int rc = L.setup(s, (const uint8_t *)tox, raw ? 1 : 0, &pk, &npk, &vk, &nvk);
    free(o1); free(o2);
    if (rc != 0) return throw_msg(env, L.last_error ? L.last_error() : "ZkpSetup failed");
    napi_value out, bpk, bvk;
    void *c1, *c2;
    napi_status s1 = napi_create_buffer_copy(env, (size_t)npk, pk, &c1, &bpk);
    napi_status s2 = napi_create_buffer_copy(env, (size_t)nvk, vk, &c2, &bvk);
    L.free_(pk); L.free_(vk);
    CHECK(env, s1); CHECK(env, s2);
    CHECK(env, napi_create_object(env, &out));
    CHECK(env, napi_set_named_property(env, out, "pk", bpk));
    CHECK(env, napi_set_named_property(env, out, "vk", bvk));
    return out;
}

static napi_value module_init(napi_env env, napi_value exports) {
    static const struct { const char *name; napi_callback fn; } fns[] = {
        {"open", js_open}, {"initAlgorithm", js_init}, {"prove", js_prove}, {"proveBatch", js_prove_batch},
        {"proveAsync", js_prove_async}, {"proveBatchAsync", js_prove_batch_async}, {"verifyBatchAsync", js_verify_batch_async}, {"verify", js_verify},
        {"initVerifier", js_init_vk}, {"allowUnpinnedKeys", js_allow}, {"setup", js_setup},
    };
    for (size_t i = 0; i < sizeof fns / sizeof fns[0]; i++) {
        napi_value f;
        if (napi_create_function(env, fns[i].name, NAPI_AUTO_LENGTH, fns[i].fn, NULL, &f) != napi_ok) return NULL;
        if (napi_set_named_property(env, exports, fns[i].name, f) != napi_ok) return NULL;
    }
    return exports;
}
NAPI_MODULE(NODE_GYP_MODULE_NAME, module_init)
