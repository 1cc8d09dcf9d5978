// gcn_asm.cpp -- TEST INFRASTRUCTURE ONLY: see gcn_asm.h.
#include "gcn_asm.h"

#include <functional>
#include <mutex>

namespace gcn {

State &state() {
    static State s;
    return s;
}

namespace {

enum Op {
    V_MAD_U64_U32, V_MAD_I64_I32, V_ADD_CO_U32, V_ADDC_CO_U32, V_SUB_CO_U32, V_SUBB_CO_U32, V_CNDMASK_B32, V_ADD_U32,
    V_LSHLREV_B32, V_LSHRREV_B32, V_LSHLREV_B64, V_LSHRREV_B64, V_ALIGNBIT_B32, V_MOV_B32, S_NOP, N_OPS
};
// operand roles: D / d = 64 / 32-bit VGPR destination, C = carry-out SGPR, c = carry-in or select SGPR, a / A = 32 / 64-bit
// source, i = immediate only
struct OpDesc {
    const char *name;
    const char *roles;
};
const OpDesc OPS[N_OPS] = {
    {"v_mad_u64_u32", "DCaaA"}, {"v_mad_i64_i32", "DCaaA"}, {"v_add_co_u32", "dCaa"},   {"v_addc_co_u32", "dCaac"},
    {"v_sub_co_u32", "dCaa"},   {"v_subb_co_u32", "dCaac"}, {"v_cndmask_b32", "daac"},  {"v_add_u32", "daa"},
    {"v_lshlrev_b32", "daa"},   {"v_lshrrev_b32", "daa"},   {"v_lshlrev_b64", "DaA"},   {"v_lshrrev_b64", "DaA"},
    {"v_alignbit_b32", "daaa"}, {"v_mov_b32", "da"},        {"s_nop", "i"},
};

enum Kind { K_ARG, K_VGPR, K_SGPR, K_IMM };
struct Opnd {
    Kind kind;
    int idx;    // argument index or first register
    int width;  // registers (1 or 2) for physical operands
    long long imm;
};
struct Ins {
    Op op;
    int n;
    Opnd o[5];
    std::string text;
};
struct Prog {
    std::vector<Ins> ins;
    size_t n_outs = 0, n_args = 0;
    std::vector<bool> vclob = std::vector<bool>(256, false), sclob = std::vector<bool>(128, false);
    bool ok = true;
};

void fail(const std::string &msg) {
    State &s = state();
    if (!s.errors++) s.first_error = msg;
    if (s.errors <= 8 && !s.quiet) fprintf(stderr, "[gcn_asm] %s\n", msg.c_str());
}

std::string trim(const std::string &x) {
    size_t a = x.find_first_not_of(" \t\r"), b = x.find_last_not_of(" \t\r");
    return a == std::string::npos ? std::string() : x.substr(a, b - a + 1);
}

bool parse_reg(const std::string &t, char file, int *idx, int *width) {
    if (t.size() < 2 || t[0] != file) return false;
    if (t[1] == '[') {
        int a, b;
        if (sscanf(t.c_str() + 2, "%d:%d]", &a, &b) != 2 || b < a) return false;
        *idx = a, *width = b - a + 1;
        return true;
    }
    char *end = nullptr;
    long v = strtol(t.c_str() + 1, &end, 10);
    if (end == t.c_str() + 1 || *end) return false;
    *idx = (int)v, *width = 1;
    return true;
}

Prog decode(const char *tmpl, const std::vector<Arg> &args, size_t n_outs, std::initializer_list<const char *> clobbers) {
    Prog p;
    p.n_outs = n_outs;
    p.n_args = args.size();
    auto bad = [&](const std::string &m) {
        p.ok = false;
        fail(m);
    };
    for (const char *c : clobbers) {
        int idx, w;
        std::string t(c);
        if (parse_reg(t, 'v', &idx, &w) && idx + w <= 256) {
            for (int k = 0; k < w; ++k) p.vclob[idx + k] = true;
        } else if (parse_reg(t, 's', &idx, &w) && idx + w <= 128) {
            for (int k = 0; k < w; ++k) p.sclob[idx + k] = true;
        } else {
            bad("unknown clobber '" + t + "'");
        }
    }
    std::string all(tmpl);
    size_t pos = 0;
    while (pos <= all.size()) {
        size_t nl = all.find('\n', pos);
        std::string line = trim(all.substr(pos, nl == std::string::npos ? std::string::npos : nl - pos));
        pos = nl == std::string::npos ? all.size() + 1 : nl + 1;
        if (line.empty()) continue;
        size_t sp = line.find_first_of(" \t");
        std::string mn = line.substr(0, sp), rest = sp == std::string::npos ? "" : line.substr(sp + 1);
        Ins in{};
        in.text = line;
        int op = -1;
        for (int k = 0; k < N_OPS; ++k)
            if (mn == OPS[k].name) op = k;
        if (op < 0) {
            bad("instruction not modelled: '" + line + "'");
            continue;
        }
        in.op = (Op)op;
        std::vector<std::string> toks;
        size_t q = 0;
        while (q <= rest.size() && !trim(rest).empty()) {
            size_t cm = rest.find(',', q);
            toks.push_back(trim(rest.substr(q, cm == std::string::npos ? std::string::npos : cm - q)));
            if (cm == std::string::npos) break;
            q = cm + 1;
        }
        const char *roles = OPS[op].roles;
        if (toks.size() != strlen(roles)) {
            bad("operand count of '" + line + "'");
            continue;
        }
        in.n = (int)toks.size();
        for (int k = 0; k < in.n; ++k) {
            const std::string &t = toks[k];
            Opnd &o = in.o[k];
            o = Opnd{K_IMM, 0, 1, 0};
            if (t.size() > 1 && t[0] == '%') {
                o.kind = K_ARG;
                int found = -1;
                if (t[1] == '[') {
                    std::string nm = t.substr(1);  // "[name]"
                    for (size_t a = 0; a < args.size(); ++a)
                        if (nm == args[a].name) found = (int)a;
                } else {
                    char *end = nullptr;
                    long v = strtol(t.c_str() + 1, &end, 10);
                    if (end != t.c_str() + 1 && !*end && v >= 0 && (size_t)v < args.size()) found = (int)v;
                }
                if (found < 0) {
                    bad("unknown operand '" + t + "' in '" + line + "'");
                    found = 0;
                }
                o.idx = found;
            } else if (parse_reg(t, 'v', &o.idx, &o.width)) {
                o.kind = K_VGPR;
                if (o.idx + o.width > 256 || o.width > 2) bad("register range in '" + line + "'");
            } else if (parse_reg(t, 's', &o.idx, &o.width)) {
                o.kind = K_SGPR;
                if (o.idx + o.width > 128 || o.width > 2) bad("register range in '" + line + "'");
            } else {
                char *end = nullptr;
                o.imm = strtoll(t.c_str(), &end, 0);
                if (end == t.c_str() || *end) bad("operand '" + t + "' in '" + line + "'");
            }
            const char r = roles[k];
            if (r == 'i' && o.kind != K_IMM) bad("immediate expected in '" + line + "'");
            if ((r == 'D' || r == 'd') && o.kind == K_ARG && (size_t)o.idx >= n_outs) bad("'" + line + "' writes an input operand");
            if ((r == 'D' || r == 'd') && (o.kind == K_SGPR || o.kind == K_IMM)) bad("VGPR destination expected in '" + line + "'");
            if ((r == 'C') && !(o.kind == K_SGPR || (o.kind == K_ARG && strchr(args[o.idx].constraint, 's'))))
                bad("SGPR carry-out expected in '" + line + "'");
            if ((r == 'C') && o.kind == K_ARG && (size_t)o.idx >= n_outs) bad("'" + line + "' writes an input operand (carry)");
            if ((r == 'c') && !(o.kind == K_SGPR || (o.kind == K_ARG && strchr(args[o.idx].constraint, 's'))))
                bad("SGPR carry-in expected in '" + line + "'");
            if (r == 'D' && ((o.kind == K_VGPR && o.width != 2) || (o.kind == K_ARG && args[o.idx].bits != 64)))
                bad("64-bit destination expected in '" + line + "'");
            if (r == 'd' && ((o.kind == K_VGPR && o.width != 1) || (o.kind == K_ARG && args[o.idx].bits != 32)))
                bad("32-bit destination expected in '" + line + "'");
            if (r == 'A' && ((o.kind != K_IMM && o.kind != K_ARG && o.width != 2) || (o.kind == K_ARG && args[o.idx].bits != 64)))
                bad("64-bit source expected in '" + line + "'");
            if (r == 'a' && ((o.kind != K_IMM && o.kind != K_ARG && o.width != 1) ||
                             (o.kind == K_ARG && args[o.idx].bits != 32 && !strchr(args[o.idx].constraint, 'n'))))
                bad("32-bit source expected in '" + line + "'");
            if ((o.kind == K_IMM) && r != 'i' && (o.imm < -16 || o.imm > 64)) bad("not an inline constant in '" + line + "'");
            if ((r == 'D' || r == 'd') && o.kind == K_VGPR)
                for (int w = 0; w < o.width; ++w)
                    if (!p.vclob[o.idx + w]) bad("'" + line + "' writes v" + std::to_string(o.idx + w) + " which is not a clobber");
            if (r == 'C' && o.kind == K_SGPR)
                for (int w = 0; w < o.width; ++w)
                    if (!p.sclob[o.idx + w]) bad("'" + line + "' writes s" + std::to_string(o.idx + w) + " which is not a clobber");
        }
        p.ins.push_back(in);
    }
    // ---- static hazard check: a VALU write of an SGPR needs two wait states before a VALU reads that SGPR (gfx940+) ----
    std::map<int, long> last_write;  // key: physical sgpr index, or 1000 + argument index
    long t = 0;
    for (const Ins &in : p.ins) {
        if (in.op == S_NOP) {
            t += in.o[0].imm + 1;
            continue;
        }
        const char *roles = OPS[in.op].roles;
        for (int k = 0; k < in.n; ++k) {
            const Opnd &o = in.o[k];
            const bool sg = o.kind == K_SGPR || (o.kind == K_ARG && strchr(args[o.idx].constraint, 's'));
            if (!sg || roles[k] == 'C') continue;
            for (int w = 0; w < (o.kind == K_SGPR ? o.width : 1); ++w) {
                auto it = last_write.find(o.kind == K_SGPR ? o.idx + w : 1000 + o.idx);
                if (it != last_write.end() && t - it->second - 1 < 2)
                    bad("hazard: '" + in.text + "' reads an SGPR a VALU wrote " + std::to_string(t - it->second - 1) +
                        " wait state(s) earlier (2 needed on gfx940+)");
            }
        }
        for (int k = 0; k < in.n; ++k)
            if (roles[k] == 'C') {
                const Opnd &o = in.o[k];
                if (o.kind == K_SGPR)
                    for (int w = 0; w < o.width; ++w) last_write[o.idx + w] = t;
                else
                    last_write[1000 + o.idx] = t;
            }
        t += 1;
    }
    return p;
}

struct Machine {
    uint32_t v[256], s[128];
    bool vdef[256], sdef[128];
    uint64_t val[24];
    bool def[24];
};

}  // namespace

void run(const char *tmpl, std::initializer_list<Arg> outs, std::initializer_list<Arg> ins,
         std::initializer_list<const char *> clobbers) {
    static std::map<const char *, Prog> cache;
    static std::mutex mu;
    std::vector<Arg> args(outs);
    args.insert(args.end(), ins.begin(), ins.end());
    const Prog *prog;
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = cache.find(tmpl);
        if (it == cache.end()) it = cache.emplace(tmpl, decode(tmpl, args, outs.size(), clobbers)).first;
        prog = &it->second;
    }
    State &st = state();
    if (prog->n_args != args.size() || args.size() > 24) {
        fail("argument list of an asm block changed between calls (or is too long)");
        return;
    }
    ++st.blocks;
    Machine m;
    memset(m.vdef, 0, sizeof m.vdef);
    memset(m.sdef, 0, sizeof m.sdef);
    for (size_t a = 0; a < args.size(); ++a) {
        const Arg &g = args[a];
        if (g.ptr) {  // output: "=..." starts undefined, "+..." carries its value in
            m.def[a] = g.constraint[0] == '+';
            m.val[a] = g.bits == 64 ? *(uint64_t *)g.ptr : *(uint32_t *)g.ptr;
        } else {
            m.def[a] = true;
            m.val[a] = g.bits == 64 ? g.value : (uint32_t)g.value;
        }
    }
    auto rd32 = [&](const Opnd &o, const Ins &in) -> uint32_t {
        switch (o.kind) {
            case K_IMM: return (uint32_t)(int32_t)o.imm;
            case K_ARG:
                if (!m.def[o.idx]) fail("'" + in.text + "' reads an operand before it is defined");
                return (uint32_t)m.val[o.idx];
            case K_VGPR:
                if (!m.vdef[o.idx]) fail("'" + in.text + "' reads v" + std::to_string(o.idx) + " before the block wrote it");
                return m.v[o.idx];
            default:
                if (!m.sdef[o.idx]) fail("'" + in.text + "' reads s" + std::to_string(o.idx) + " before the block wrote it");
                return m.s[o.idx];
        }
    };
    auto rd64 = [&](const Opnd &o, const Ins &in) -> uint64_t {
        switch (o.kind) {
            case K_IMM: return (uint64_t)(int64_t)o.imm;
            case K_ARG:
                if (!m.def[o.idx]) fail("'" + in.text + "' reads an operand before it is defined");
                return m.val[o.idx];
            case K_VGPR:
                if (!m.vdef[o.idx] || !m.vdef[o.idx + 1]) fail("'" + in.text + "' reads v[" + std::to_string(o.idx) + ":] before the block wrote it");
                return (uint64_t)m.v[o.idx] | ((uint64_t)m.v[o.idx + 1] << 32);
            default:
                if (!m.sdef[o.idx] || !m.sdef[o.idx + 1]) fail("'" + in.text + "' reads s[" + std::to_string(o.idx) + ":] before the block wrote it");
                return (uint64_t)m.s[o.idx] | ((uint64_t)m.s[o.idx + 1] << 32);
        }
    };
    auto rdc = [&](const Opnd &o, const Ins &in) -> uint32_t {  // this lane's bit of a carry / select mask
        if (o.kind == K_ARG) {
            if (!m.def[o.idx]) fail("'" + in.text + "' reads a carry before it is defined");
            return (uint32_t)(m.val[o.idx] & 1);
        }
        if (!m.sdef[o.idx]) fail("'" + in.text + "' reads s" + std::to_string(o.idx) + " before the block wrote it");
        return m.s[o.idx] & 1;
    };
    auto wr32 = [&](const Opnd &o, uint32_t x) {
        if (o.kind == K_ARG) m.val[o.idx] = x, m.def[o.idx] = true;
        else m.v[o.idx] = x, m.vdef[o.idx] = true;
    };
    auto wr64 = [&](const Opnd &o, uint64_t x) {
        if (o.kind == K_ARG) {
            m.val[o.idx] = x, m.def[o.idx] = true;
        } else {
            m.v[o.idx] = (uint32_t)x, m.v[o.idx + 1] = (uint32_t)(x >> 32);
            m.vdef[o.idx] = m.vdef[o.idx + 1] = true;
        }
    };
    auto wrc = [&](const Opnd &o, uint32_t bit) {
        if (o.kind == K_ARG) {
            m.val[o.idx] = bit, m.def[o.idx] = true;
        } else {
            m.s[o.idx] = bit, m.sdef[o.idx] = true;
            if (o.width == 2) m.s[o.idx + 1] = 0, m.sdef[o.idx + 1] = true;
        }
    };
    for (const Ins &in : prog->ins) {
        const Opnd *o = in.o;
        switch (in.op) {
            case V_MAD_U64_U32: {
                const uint64_t a = rd32(o[2], in), b = rd32(o[3], in), c = rd64(o[4], in);
                const unsigned __int128 r = (unsigned __int128)(a * b) + c;
                wr64(o[0], (uint64_t)r);
                wrc(o[1], (uint32_t)(r >> 64) & 1);
                break;
            }
            case V_MAD_I64_I32: {
                const int64_t a = (int32_t)rd32(o[2], in), b = (int32_t)rd32(o[3], in);
                const uint64_t c = rd64(o[4], in);
                const uint64_t r = (uint64_t)(a * b) + c;
                wr64(o[0], r);
                // the carry-out of the signed form is the overflow bit; the product never consumes it
                const __int128 full = (__int128)(a * b) + (__int128)(int64_t)c;
                wrc(o[1], full != (__int128)(int64_t)r);
                break;
            }
            case V_ADD_CO_U32: {
                const uint64_t r = (uint64_t)rd32(o[2], in) + rd32(o[3], in);
                wr32(o[0], (uint32_t)r);
                wrc(o[1], (uint32_t)(r >> 32));
                break;
            }
            case V_ADDC_CO_U32: {
                const uint64_t r = (uint64_t)rd32(o[2], in) + rd32(o[3], in) + rdc(o[4], in);
                wr32(o[0], (uint32_t)r);
                wrc(o[1], (uint32_t)(r >> 32));
                break;
            }
            case V_SUB_CO_U32: {
                const uint64_t a = rd32(o[2], in), b = rd32(o[3], in);
                wr32(o[0], (uint32_t)(a - b));
                wrc(o[1], a < b);
                break;
            }
            case V_SUBB_CO_U32: {
                const uint64_t a = rd32(o[2], in), b = (uint64_t)rd32(o[3], in) + rdc(o[4], in);
                wr32(o[0], (uint32_t)(a - b));
                wrc(o[1], a < b);
                break;
            }
            case V_CNDMASK_B32: {
                const uint32_t a = rd32(o[1], in), b = rd32(o[2], in);
                wr32(o[0], rdc(o[3], in) ? b : a);
                break;
            }
            case V_ADD_U32: wr32(o[0], rd32(o[1], in) + rd32(o[2], in)); break;
            case V_LSHLREV_B32: {
                const uint32_t sh = rd32(o[1], in) & 31;
                wr32(o[0], rd32(o[2], in) << sh);
                break;
            }
            case V_LSHRREV_B32: {
                const uint32_t sh = rd32(o[1], in) & 31;
                wr32(o[0], rd32(o[2], in) >> sh);
                break;
            }
            case V_LSHLREV_B64: {
                const uint32_t sh = rd32(o[1], in) & 63;
                wr64(o[0], rd64(o[2], in) << sh);
                break;
            }
            case V_LSHRREV_B64: {
                const uint32_t sh = rd32(o[1], in) & 63;
                wr64(o[0], rd64(o[2], in) >> sh);
                break;
            }
            case V_ALIGNBIT_B32: {  // ({S0, S1} >> S2[4:0]) & 0xffffffff
                const uint64_t cat = ((uint64_t)rd32(o[1], in) << 32) | rd32(o[2], in);
                wr32(o[0], (uint32_t)(cat >> (rd32(o[3], in) & 31)));
                break;
            }
            case V_MOV_B32: wr32(o[0], rd32(o[1], in)); break;
            case S_NOP: break;
            default: break;
        }
        ++st.executed;
    }
    size_t a = 0;
    for (const Arg &g : outs) {
        if (!m.def[a]) fail(std::string("output operand ") + g.name + " was never written");
        if (g.bits == 64) *(uint64_t *)g.ptr = m.val[a];
        else *(uint32_t *)g.ptr = (uint32_t)m.val[a];
        ++a;
    }
}

}  // namespace gcn

// exported from the emulator library only (tests/emu_backend.py): switch the interpreter, read its counters
extern "C" int p2hot_emu_asm(int on) {
    const int was = gcn::state().on;
    gcn::state().on = on != 0;
    return was;
}
extern "C" unsigned long long p2hot_emu_asm_stats(unsigned long long *blocks, unsigned long long *errors, char *first_error,
                                                  size_t cap) {
    gcn::State &s = gcn::state();
    if (blocks) *blocks = s.blocks;
    if (errors) *errors = s.errors;
    if (first_error && cap) {
        strncpy(first_error, s.first_error.c_str(), cap - 1);
        first_error[cap - 1] = 0;
    }
    return s.executed;
}

// The checker checked: each deliberately broken block must be reported, the sound one must not.  Returns a bit per case
// that behaved as expected (0x7f = all); the interpreter's error state is restored afterwards.
extern "C" unsigned p2hot_emu_asm_negative_tests() {
    gcn::State saved = gcn::state();
    unsigned okmask = 0;
    uint32_t a = 0xFFFFFFFFu, b = 2, r = 0, r2 = 0;
    auto errors_of = [&](const std::function<void()> &f) {
        gcn::state().errors = 0;
        gcn::state().first_error.clear();
        gcn::state().quiet = true;  // the expected reports are noise
        f();
        gcn::state().quiet = false;
        return gcn::state().errors;
    };
    // 0: sound block: {r2, r} = a + b as a 64-bit sum, two wait states before the carry is read
    if (errors_of([&] {
            gcn::run("v_add_co_u32 %[r], s[40:41], %[a], %[b]\n\ts_nop 1\n\tv_addc_co_u32 %[r2], s[40:41], 0, 0, s[40:41]",
                     {gcn::out("[r]", "=&v", r), gcn::out("[r2]", "=&v", r2)}, {gcn::in("[a]", "v", a), gcn::in("[b]", "v", b)},
                     {"s40", "s41"});
        }) == 0 && r == 1 && r2 == 1)
        okmask |= 1;
    // 1: the carry read one wait state after the VALU wrote it
    if (errors_of([&] {
            gcn::run("v_add_co_u32 %[r], s[40:41], %[a], %[b]\n\ts_nop 0\n\tv_addc_co_u32 %[r2], s[40:41], 0, 0, s[40:41]",
                     {gcn::out("[r]", "=&v", r), gcn::out("[r2]", "=&v", r2)}, {gcn::in("[a]", "v", a), gcn::in("[b]", "v", b)},
                     {"s40", "s41"});
        }) > 0)
        okmask |= 2;
    // 2: a physical register written without being declared a clobber
    if (errors_of([&] {
            gcn::run("v_add_u32 v72, %[a], %[b]\n\tv_add_u32 %[r], v72, 0", {gcn::out("[r]", "=&v", r)},
                     {gcn::in("[a]", "v", a), gcn::in("[b]", "v", b)}, {"v71"});
        }) > 0)
        okmask |= 4;
    // 3: a scratch register read before the block wrote it
    if (errors_of([&] {
            gcn::run("v_add_u32 %[r], v70, %[b]", {gcn::out("[r]", "=&v", r)}, {gcn::in("[b]", "v", b)}, {"v70"});
        }) > 0)
        okmask |= 8;
    // 4: an instruction the interpreter does not model
    if (errors_of([&] {
            gcn::run("v_mul_lo_u32 %[r], %[a], %[b]", {gcn::out("[r]", "=&v", r)}, {gcn::in("[a]", "v", a), gcn::in("[b]", "v", b)}, {});
        }) > 0)
        okmask |= 16;
    // 5: an input operand used as a destination
    if (errors_of([&] {
            gcn::run("v_add_u32 %[a], %[a], %[b]\n\tv_add_u32 %[r], %[a], 0", {gcn::out("[r]", "=&v", r)},
                     {gcn::in("[a]", "v", a), gcn::in("[b]", "v", b)}, {});
        }) > 0)
        okmask |= 32;
    // 6: an output that is never written
    if (errors_of([&] {
            gcn::run("v_add_u32 %[r], %[a], %[b]", {gcn::out("[r]", "=&v", r), gcn::out("[r2]", "=&v", r2)},
                     {gcn::in("[a]", "v", a), gcn::in("[b]", "v", b)}, {});
        }) > 0)
        okmask |= 64;
    gcn::state() = saved;
    return okmask;
}
