/*
 * tables.cpp — access to the numeric tables blob (data/saf_tables.bin, produced by
 * tools/extract_tables.py and linked into the library with `ld -r -b binary`).
 */
#include "saf_hip_common.h"
#include <cstdint>

extern "C" const unsigned char _binary_saf_tables_bin_start[];
extern "C" const unsigned char _binary_saf_tables_bin_end[];

namespace saf {

struct Tab { std::string name; int d0, d1; std::vector<float> data; };
static std::vector<Tab>* g_tabs = nullptr;

static void parse()
{
    if (g_tabs) return;
    g_tabs = new std::vector<Tab>();
    const unsigned char* p = _binary_saf_tables_bin_start;
    const unsigned char* e = _binary_saf_tables_bin_end;
    if (e - p < 12 || memcmp(p, "SAFT", 4) != 0) SAF_FATAL("embedded tables blob is corrupt");
    uint32_t ver, n;
    memcpy(&ver, p + 4, 4); memcpy(&n, p + 8, 4); p += 12;
    for (uint32_t i = 0; i < n; i++) {
        uint32_t nl, d0, d1;
        memcpy(&nl, p, 4); p += 4;
        Tab t; t.name.assign((const char*)p, nl); p += nl;
        memcpy(&d0, p, 4); memcpy(&d1, p + 4, 4); p += 8;
        t.d0 = (int)d0; t.d1 = (int)d1;
        t.data.resize((size_t)d0 * d1);
        memcpy(t.data.data(), p, (size_t)d0 * d1 * 4); p += (size_t)d0 * d1 * 4;
        g_tabs->push_back(std::move(t));
    }
}

const float* table(const char* name, int* d0, int* d1)
{
    parse();
    for (auto& t : *g_tabs)
        if (t.name == name) { if (d0) *d0 = t.d0; if (d1) *d1 = t.d1; return t.data.data(); }
    return nullptr;
}

const float* table_required(const char* name, int count)
{
    int d0 = 0, d1 = 0;
    const float* p = table(name, &d0, &d1);
    if (!p || d0 * d1 != count) SAF_FATAL("table '%s' missing or of unexpected size (%d x %d, wanted %d)", name, d0, d1, count);
    return p;
}

}  // namespace saf
