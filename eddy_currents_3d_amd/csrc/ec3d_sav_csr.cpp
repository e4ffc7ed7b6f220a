// ec3d_sav_csr.cpp — recognise the structured A-V form (MatView::sav) in a CSR matrix.
//
// The drop-in entry points receive the matrix the reference assembled (src/EC3D.f90:465-1049) as plain
// CSR.  Its structure is always the same: three copies of a 7-point operator on an sdx*sdy*sdz grid
// (Ax, Ay, Az), rows of conducting cells extended by 2-3 couplings to the scalar potential U, and one U
// row per conducting cell (numbered in scan order) with 7 U neighbours and up to 6 A couplings.  This file
// checks that structure entry by entry and, when every entry fits a stencil slot and the stored order of
// each row is the order the kernels add the slots in, produces the class-coded form: one class byte per
// row and a table of 16 coefficients per class.  Nothing is assumed about the VALUES (they become the
// class table as they are, bit for bit); any row that does not fit makes the caller fall back to the
// general bands + tail format.  Returns 0 on success, -1 when the form does not apply.
#include "ec3d_internal.hpp"

#include <algorithm>
#include <cstring>
#include <string>
#include <unordered_map>

namespace {
struct Key16 {
    uint64_t w[16];
    bool operator==(const Key16 &o) const { return memcmp(w, o.w, sizeof w) == 0; }
};
struct Key16Hash {
    size_t operator()(const Key16 &k) const
    {
        uint64_t h = 0x9E3779B97F4A7C15ull;
        for (int i = 0; i < 16; ++i) h = (h ^ k.w[i]) * 0xBF58476D1CE4E5B9ull + (h >> 29);
        return (size_t)h;
    }
};
int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }
} // namespace

int ec3d_csr_to_sav_host(int64_t n, const double *valA, const int32_t *irow, const int32_t *jcol, SavHost &S)
{
    if (n <= 0 || irow[0] != 1) return -1;
    const int64_t nnz = (int64_t)irow[n] - 1;
    // 1. the grid: offsets carried by >= 40 % of a row sample must be {-kdz, -sdx, -1, 0, 1, sdx, kdz}
    int64_t sdx = 0, plane = 0;
    {
        const int64_t stride = std::max<int64_t>(1, n / (1 << 18));
        std::unordered_map<int64_t, int64_t> cnt;
        int64_t rows = 0;
        for (int64_t r = 0; r < n; r += stride, ++rows)
            for (int64_t p = irow[r] - 1; p < irow[r + 1] - 1; ++p) ++cnt[(int64_t)jcol[p] - 1 - r];
        std::vector<int64_t> offs;
        for (auto &kv : cnt)
            if (kv.second * 10 >= rows * 4) offs.push_back(kv.first);
        std::sort(offs.begin(), offs.end());
        if (offs.size() != 7 || offs[3] != 0 || offs[4] != 1 || offs[2] != -1 || offs[1] != -offs[5] ||
            offs[0] != -offs[6])
            return -1;
        sdx = offs[5];
        plane = offs[6];
    }
    if (sdx < 5 || plane % sdx != 0 || plane / sdx < 5) return -1;
    // 2. the A block ends at the first row that reaches further back than one plane (a U row's Ax column)
    int64_t nA = n;
    for (int64_t r = 0; r < n && nA == n; ++r)
        for (int64_t p = irow[r] - 1; p < irow[r + 1] - 1; ++p) {
            const int64_t col = (int64_t)jcol[p] - 1;
            if (col < 0 || col >= n) return -1;
            if (col < r - plane) { nA = r; break; }
        }
    if (nA % 3 != 0) return -1;
    const int64_t nC = nA / 3, nU = n - nA;
    if (nC % plane != 0 || nC / plane < 5) return -1;
    const int64_t sdz = nC / plane;
    // 3. conducting cells = cells whose Ax row has a U column; U(m) must be the m-th of them
    std::vector<int32_t> ucell;
    ucell.reserve((size_t)nU);
    for (int64_t q = 0; q < nC; ++q) {
        bool cond = false;
        for (int64_t p = irow[q] - 1; p < irow[q + 1] - 1 && !cond; ++p) cond = (int64_t)jcol[p] - 1 >= nA;
        if (cond) {
            if ((int64_t)ucell.size() == nU) return -1;
            ucell.push_back((int32_t)q);
        }
    }
    if ((int64_t)ucell.size() != nU) return -1;

    // device layout (see ec3d_ctx::pitch): tile-aligned planes when that costs < 1/16 in rows
    int64_t pitch = plane;
    {
        const int64_t pp = round_up(plane, EC3D_TILE);
        bool want = (pp - plane) * 16 <= plane && sdz >= 8;
        if (const char *e = getenv("EC3D_PITCH")) want = atoi(e) == 2 || (want && atoi(e) != 0);
        if (want) pitch = pp;
    }
    const int64_t nCd = pitch * sdz, n_dev = 4 * nCd;
    if (n_dev > (int64_t)INT32_MAX - EC3D_TILE) return -1;
    auto dev_cell = [&](int64_t q) { return (q / plane) * pitch + q % plane; };
    auto dev_of = [&](int64_t ref) { // device row of a reference unknown
        if (ref < nA) return (ref / nC) * nCd + dev_cell(ref % nC);
        return 3 * nCd + dev_cell(ucell[(size_t)(ref - nA)]);
    };
    const int64_t boff[7] = {-pitch, -sdx, -1, 0, 1, sdx, pitch}, step[3] = {1, sdx, pitch};
    auto band_of = [&](int64_t d) {
        for (int b = 0; b < 7; ++b)
            if (boff[b] == d) return b;
        return -1;
    };

    // 4. every row -> 16 slot coefficients; the kernels add slots 7..15 then 0..6 (U rows) or 0..6 then
    //    7..11 (A rows): the stored order must be that order
    S = SavHost();
    S.n_pad = round_up(n_dev, EC3D_TILE);
    S.cls.assign((size_t)S.n_pad, 0);
    S.tile_flag.assign((size_t)(S.n_pad / EC3D_TILE), 0);
    std::unordered_map<Key16, int, Key16Hash> dict[3]; // kind 0: plain A row, 1: A row with U columns, 2: U row
    std::vector<Key16> keys[3];
    std::vector<uint8_t> kind((size_t)n);
    std::vector<int32_t> local((size_t)n);
    for (int64_t r = 0; r < n; ++r) {
        const int64_t dr = dev_of(r);
        const bool urow = r >= nA;
        const int d = urow ? 3 : (int)(r / nC);
        double t[16];
        for (double &v : t) v = 0.0;
        int last = -1; // evaluation index of the previous entry
        bool coupled = false;
        for (int64_t p = irow[r] - 1; p < irow[r + 1] - 1; ++p) {
            const int64_t col = (int64_t)jcol[p] - 1;
            if (col < 0 || col >= n) return -1;
            const int64_t dc = dev_of(col);
            int slot, ev;
            if (!urow) {
                if (col < nA) {
                    slot = band_of(dc - dr);
                    ev = slot;
                } else {
                    const int64_t diff = dc - (dr + (3 - d) * nCd);
                    if (diff % step[d] != 0) return -1;
                    const int64_t m = diff / step[d];
                    if (m < -2 || m > 2) return -1;
                    slot = 7 + (int)m + 2;
                    ev = slot;
                    coupled = true;
                }
            } else {
                if (col < nA) {
                    const int dd = (int)(dc / nCd);
                    const int64_t diff = dc - (dr - (3 - dd) * nCd);
                    if (diff % step[dd] != 0) return -1;
                    const int64_t j = diff / step[dd];
                    if (j < -1 || j > 1) return -1;
                    slot = 7 + 3 * dd + (int)j + 1;
                    ev = slot - 7;
                } else {
                    slot = band_of(dc - dr);
                    ev = 9 + slot;
                }
            }
            if (slot < 0 || ev <= last) return -1;
            last = ev;
            t[slot] = valA[p];
        }
        Key16 k;
        memcpy(k.w, t, sizeof t);
        const int kd = urow ? 2 : (coupled ? 1 : 0);
        auto it = dict[kd].find(k);
        int id;
        if (it == dict[kd].end()) {
            id = (int)keys[kd].size();
            if (keys[0].size() + keys[1].size() + keys[2].size() >= 255) return -1;
            dict[kd].emplace(k, id);
            keys[kd].push_back(k);
        } else {
            id = it->second;
        }
        kind[(size_t)r] = (uint8_t)kd;
        local[(size_t)r] = id;
        if (kd) S.tile_flag[(size_t)(dr / EC3D_TILE)] = 1;
    }
    S.a0 = (int)keys[0].size();
    S.u0 = S.a0 + (int)keys[1].size();
    S.zero = S.u0 + (int)keys[2].size();
    S.ncls = S.zero + 1;
    S.table.assign((size_t)S.ncls * 16, 0.0);
    const int base[3] = {0, S.a0, S.u0};
    for (int kd = 0; kd < 3; ++kd)
        for (size_t i = 0; i < keys[kd].size(); ++i)
            memcpy(&S.table[((size_t)base[kd] + i) * 16], keys[kd][i].w, 16 * sizeof(double));
    std::fill(S.cls.begin(), S.cls.end(), (uint8_t)S.zero);
    for (int64_t r = 0; r < n; ++r) S.cls[(size_t)dev_of(r)] = (uint8_t)(base[kind[(size_t)r]] + local[(size_t)r]);
    // 5. U block: only tiles that hold an unknown are visited
    S.ntiles_front = (3 * nCd + EC3D_TILE - 1) / EC3D_TILE;
    for (int64_t tl = S.ntiles_front; tl < S.n_pad / EC3D_TILE; ++tl)
        if (S.tile_flag[(size_t)tl]) S.ulist.push_back((int32_t)tl);
    S.cond_cell.resize((size_t)nU);
    for (int64_t m = 0; m < nU; ++m) S.cond_cell[(size_t)m] = (int32_t)dev_cell(ucell[(size_t)m]);
    S.n_ref = n;
    S.nnz = nnz;
    S.sdx = sdx;
    S.plane = plane;
    S.pitch = pitch;
    S.nCd = nCd;
    S.n_dev = n_dev;
    return 0;
}

// The four blocks are cut into slabs one by one, so nothing may couple ACROSS a block boundary through the
// +-plane bands: true for the reference's system (the first and last plane of a component are box-boundary
// rows, src/EC3D.f90:528-646), not for e.g. a single-component cube that the recogniser reads as three
// "blocks" of sdz/3 planes -- on one GPU that reading is harmless (bands simply run on), cut into slabs it is not.
int ec3d_sav_cuttable(const SavHost &G, int nranks, std::string &why)
{
    const int64_t sdz = G.nCd / G.pitch;
    if (nranks > 1 && sdz < 2 * (int64_t)nranks) {
        why = "every rank needs at least two z-planes";
        return 2;
    }
    if (nranks > 1)
        for (int d = 0; d < 4; ++d)
            for (int side = 0; side < 2; ++side) {
                const int64_t pl = side ? sdz - 1 : 0, base = d * G.nCd + pl * G.pitch;
                for (int64_t q = 0; q < G.plane; ++q)
                    if (G.table[(size_t)G.cls[(size_t)(base + q)] * 16 + (side ? 6 : 0)] != 0.0) {
                        why = "the matrix couples across what would be the z faces of a component (not the reference's "
                              "A-V system on a box); use one GPU";
                        return 7;
                    }
            }
    return 0;
}

// One z-slab of a recognised system: the held planes [e0, e1) of each of the four blocks, same pitch, same
// class table.  Rows of the halo planes (outside [k0, k1)) get the all-zero class: they only carry the
// neighbours' values, exactly like the halo rows ec3d_assemble_slab produces natively.
void ec3d_sav_slice(const SavHost &G, int64_t e0, int64_t e1, int64_t k0, int64_t k1, SavHost &L)
{
    L = SavHost();
    const int64_t np = e1 - e0, pitch = G.pitch, nCd = np * pitch;
    L.sdx = G.sdx;
    L.plane = G.plane;
    L.pitch = pitch;
    L.nCd = nCd;
    L.n_dev = 4 * nCd;
    L.n_pad = round_up(L.n_dev, EC3D_TILE);
    L.a0 = G.a0; L.u0 = G.u0; L.zero = G.zero; L.ncls = G.ncls;
    L.table = G.table;
    L.cls.assign((size_t)L.n_pad, (uint8_t)G.zero);
    for (int d = 0; d < 4; ++d)
        for (int64_t p = k0; p < k1; ++p)
            memcpy(&L.cls[(size_t)(d * nCd + (p - e0) * pitch)], &G.cls[(size_t)(d * G.nCd + p * pitch)], (size_t)pitch);
    L.tile_flag.assign((size_t)(L.n_pad / EC3D_TILE), 0);
    for (int64_t r = 0; r < L.n_dev; ++r) {
        const int c = L.cls[(size_t)r];
        if (c >= L.a0 && c < L.zero) L.tile_flag[(size_t)(r / EC3D_TILE)] = 1;
    }
    L.ntiles_front = (3 * nCd + EC3D_TILE - 1) / EC3D_TILE;
    for (int64_t tl = L.ntiles_front; tl < L.n_pad / EC3D_TILE; ++tl)
        if (L.tile_flag[(size_t)tl]) L.ulist.push_back((int32_t)tl);
    // held conducting cells: scan order is plane-major, so they are one contiguous run of the global list
    const int64_t lo = e0 * pitch, hi = e1 * pitch;
    bool first = true;
    for (size_t m = 0; m < G.cond_cell.size(); ++m) {
        const int64_t cell = G.cond_cell[m];
        if (cell < lo || cell >= hi) continue;
        if (first) { L.u_first = (int64_t)m; first = false; }
        L.cond_cell.push_back((int32_t)(cell - lo));
    }
    L.n_ref = 3 * np * G.plane + (int64_t)L.cond_cell.size();
    L.nnz = 0;
    L.nown = 4;
    for (int d = 0; d < 4; ++d) {
        L.own_lo[d] = d * nCd + (k0 - e0) * pitch;
        L.own_hi[d] = d * nCd + (k1 - e0) * pitch;
    }
    L.halo = pitch;
}
