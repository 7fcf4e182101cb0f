// hr_scene.inl — a section of hr_core.hip (included there, inside its extern "C" block): geometry ingest, hr_scene_commit (build, refit,
// the tree cache), textures, materials, lights, sample tables, and what a pass needs uploaded before it starts.
static const size_t kStageBytes = (size_t)16 << 20;

// host bytes -> device through the pinned ring: while the DMA of one half runs, the CPU fills the other
static int stagedUpload(hr_ctx *c, char *dst, const char *src, size_t bytes)
{
    for (int k = 0; k < 2; ++k) {
        if (!c->stage[k]) {
            HIP_TRY(c, hipHostMalloc((void **)&c->stage[k], kStageBytes, hipHostMallocDefault));
            HIP_TRY(c, hipEventCreateWithFlags(&c->stageEv[k], hipEventDisableTiming));
        }
    }
    for (size_t at = 0; at < bytes; at += kStageBytes) {
        const size_t len = bytes - at < kStageBytes ? bytes - at : kStageBytes;
        const int k = c->stageTurn++ & 1;
        if (c->stageBusy[k]) HIP_TRY(c, hipEventSynchronize(c->stageEv[k]));
        std::memcpy(c->stage[k], src + at, len);
        HIP_TRY(c, hipMemcpyAsync(dst + at, c->stage[k], len, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipEventRecord(c->stageEv[k], c->stream));
        c->stageBusy[k] = true;
    }
    return HR_OK;
}

int hr_geom_add(hr_ctx *c, const hr_mesh_desc *d, hr_geom_id *out)
{
    ENTER(c);
    if (!d || !d->positions || !d->normals || !d->indices || d->n_vertices <= 0 || d->n_indices < 0)
        FAIL(c, HR_ERR_INVALID, "mesh needs positions, normals, indices");
    if (d->mode != HR_TRIANGLES && d->mode != HR_TRIANGLE_STRIP) FAIL(c, HR_ERR_INVALID, "unsupported draw mode");
    {
        uint32_t worst = 0; // (a plain max reduction: vectorises)
        for (int i = 0; i < d->n_indices; ++i) worst = d->indices[i] > worst ? d->indices[i] : worst;
        if (d->n_indices > 0 && worst >= (uint32_t)d->n_vertices) FAIL(c, HR_ERR_INVALID, "index out of range");
    }
    {
        // positions must be finite: a NaN box has no order, and the tree builders' progress arguments (and every slab test) assume one
        const int sb = d->position_stride == 0 ? 12 : d->position_stride;
        if (sb < 12) FAIL(c, HR_ERR_INVALID, "attribute stride smaller than the attribute");
        float worst = 0.0f;
        bool nan = false;
        for (int i = 0; i < d->n_vertices; ++i) {
            float p[3];
            std::memcpy(p, (const char *)d->positions + (size_t)i * (size_t)sb, 12);
            const float m = std::fmax(std::fabs(p[0]), std::fmax(std::fabs(p[1]), std::fabs(p[2]))); // (fmax drops a NaN operand: checked apart)
            worst = m > worst ? m : worst;
            nan = nan || p[0] != p[0] || p[1] != p[1] || p[2] != p[2];
        }
        if (nan || !(worst <= 3.0e37f)) FAIL(c, HR_ERR_INVALID, "vertex positions must be finite");
        for (int k = 0; k < 16; ++k)
            if (!(std::fabs(d->world_from_entity[k]) <= 3.0e37f)) FAIL(c, HR_ERR_INVALID, "world_from_entity must be finite");
    }
    const float *src[6] = {d->positions, d->normals, d->uvs, d->tangents, d->bitangents, d->colors};
    const int32_t strideB[6] = {d->position_stride, d->normal_stride, d->uv_stride, d->tangent_stride, d->bitangent_stride, d->color_stride};
    const int comps[6] = {3, 3, 2, 3, 3, 3};
    Geom g;
    g.alive = true;
    g.nVerts = d->n_vertices;
    g.nIdx = (uint32_t)d->n_indices;
    g.mode = d->mode;
    std::memcpy(g.world, d->world_from_entity, sizeof(g.world));
    g.frontFaceCW = d->front_face_cw, g.isOccluder = d->is_occluder, g.material = d->material_id;
    // layout of the device block: every attribute as the caller holds it (its stride included), then the indices
    size_t bytesOf[6] = {0, 0, 0, 0, 0, 0}, total = 0;
    std::vector<float> tight[6]; // only for strides that are not a multiple of four bytes (re-packed on the host)
    // One interleaved vertex buffer (every attribute a pointer into the same array of `stride`-byte vertices, as glTF loaders hand
    // them over) is uploaded ONCE and addressed with per-attribute offsets; uploading it once per attribute with its full stride
    // cost 3-6 x the device memory and PCIe traffic.
    const char *ilo = nullptr, *ihi = nullptr;
    int isb = 0;
    bool interleaved = true;
    int nAttr = 0;
    for (int a = 0; a < 6; ++a) {
        if (!src[a]) continue;
        ++nAttr;
        const char *p0 = (const char *)src[a], *p1 = p0 + comps[a] * sizeof(float);
        if (strideB[a] <= 0 || strideB[a] % 4 != 0 || (isb != 0 && strideB[a] != isb)) interleaved = false;
        isb = strideB[a];
        ilo = (!ilo || p0 < ilo) ? p0 : ilo;
        ihi = (!ihi || p1 > ihi) ? p1 : ihi;
    }
    interleaved = interleaved && nAttr >= 2 && (size_t)(ihi - ilo) <= (size_t)isb;
    if (interleaved) {
        const size_t span = (size_t)(g.nVerts - 1) * (size_t)isb + (size_t)(ihi - ilo);
        for (int a = 0; a < 6; ++a) {
            if (!src[a]) continue;
            g.has[a] = true;
            g.stride[a] = isb / 4;
            g.off[a] = (size_t)((const char *)src[a] - ilo);
            bytesOf[a] = 0;
        }
        total = (span + 15) & ~(size_t)15;
    } else {
        for (int a = 0; a < 6; ++a) {
            if (!src[a]) continue;
            g.has[a] = true;
            int sb = strideB[a] == 0 ? comps[a] * (int)sizeof(float) : strideB[a];
            if (sb < comps[a] * (int)sizeof(float) && sb != 0) FAIL(c, HR_ERR_INVALID, "attribute stride smaller than the attribute");
            if (sb % 4 != 0) {
                tight[a].resize((size_t)g.nVerts * comps[a]);
                for (int i = 0; i < g.nVerts; ++i) std::memcpy(&tight[a][(size_t)i * comps[a]], (const char *)src[a] + (size_t)i * sb, comps[a] * sizeof(float));
                sb = comps[a] * (int)sizeof(float);
            }
            g.stride[a] = sb / 4;
            bytesOf[a] = (size_t)(g.nVerts - 1) * sb + comps[a] * sizeof(float);
            g.off[a] = total;
            total += (bytesOf[a] + 15) & ~(size_t)15;
        }
    }
    g.off[6] = total;
    total += ((size_t)g.nIdx * 4 + 15) & ~(size_t)15;
    g.blockBytes = total;
    g.dBlock = c->meshAlloc(total ? total : 16, &g.chunk);
    if (!g.dBlock) FAIL(c, HR_ERR_DEVICE, "out of device memory for a mesh block");
    // (the 16-byte alignment padding behind each range is never uploaded, yet the tree cache's content hash covers the whole block:
    // recycled device memory there made the key differ from run to run)
    int rc = HR_OK;
    if (hipMemsetAsync(g.dBlock, 0, total ? total : 16, c->stream) != hipSuccess) {
        c->meshRelease(g.chunk); // (every error return behind meshAlloc gives the block back)
        FAIL(c, HR_ERR_DEVICE, "hipMemsetAsync of a mesh block failed");
    }
    if (interleaved) {
        rc = stagedUpload(c, g.dBlock, ilo, (size_t)(g.nVerts - 1) * (size_t)isb + (size_t)(ihi - ilo));
    } else {
        for (int a = 0; a < 6 && rc == HR_OK; ++a)
            if (g.has[a]) rc = stagedUpload(c, g.dBlock + g.off[a], tight[a].empty() ? (const char *)src[a] : (const char *)tight[a].data(), bytesOf[a]);
    }
    if (rc == HR_OK && g.nIdx) rc = stagedUpload(c, g.dBlock + g.off[6], (const char *)d->indices, (size_t)g.nIdx * 4);
    if (rc != HR_OK) {
        c->meshRelease(g.chunk);
        return rc;
    }
    c->geoms.push_back(g);
    c->committed = false, c->topologyDirty = true;
    if (out) *out = (hr_geom_id)c->geoms.size() - 1;
    return HR_OK;
}

int hr_geom_remove(hr_ctx *c, hr_geom_id id)
{
    ENTER(c);
    if (id < 0 || id >= (int)c->geoms.size() || !c->geoms[id].alive) FAIL(c, HR_ERR_INVALID, "bad geom id");
    for (int k = 0; k < 2; ++k) // an upload of this mesh may still be in flight (nothing to wait for when the staging ring is idle)
        if (c->stageBusy[k] && hipEventQuery(c->stageEv[k]) != hipSuccess) HIP_TRY(c, hipEventSynchronize(c->stageEv[k]));
    c->meshRelease(c->geoms[id].chunk);
    c->geoms[id] = Geom();
    c->committed = false, c->topologyDirty = true;
    return HR_OK;
}

int hr_geom_set_transform(hr_ctx *c, hr_geom_id id, const float m[16])
{
    ENTER(c);
    if (id < 0 || id >= (int)c->geoms.size() || !c->geoms[id].alive || !m) FAIL(c, HR_ERR_INVALID, "bad geom id");
    for (int k = 0; k < 16; ++k)
        if (!(std::fabs(m[k]) <= 3.0e37f)) FAIL(c, HR_ERR_INVALID, "world_from_entity must be finite");
    std::memcpy(c->geoms[id].world, m, 16 * sizeof(float));
    c->committed = false, c->transformDirty = true;
    return HR_OK;
}

int hr_scene_clear(hr_ctx *c)
{
    ENTER(c);
    for (int k = 0; k < 2; ++k)
        if (c->stageBusy[k] && hipEventQuery(c->stageEv[k]) != hipSuccess) HIP_TRY(c, hipEventSynchronize(c->stageEv[k]));
    c->meshReleaseAll();
    c->geoms.clear();
    c->committed = false, c->topologyDirty = true;
    return HR_OK;
}

// ---- tree cache file (hr_scene_cache): header + nodes + node boxes + prim -> slot map
namespace {
struct CacheHeader {
    char magic[8];
    uint32_t version, nodeBytes;
    unsigned long long key;
    uint32_t nTris, nNodes, levels, rootLeafCount, triSlots, builder; // builder: which binary tree was collapsed (BuildResult::builder)
    uint32_t levelStart[kMaxLevels + 1];
    float costRadix, costPloc;     // the candidates' costs as hr_scene_info reports them
    unsigned long long payloadSum; // checksum of everything behind the header (the key covers the SCENE, not the file)
};
const uint32_t kCacheVersion = 4; // 4: two candidate builders (round 4) — the header says which tree the file holds, the key which builder options made it

// 64-bit checksum of the payload, eight bytes at a time (the files are tens to hundreds of MB)
unsigned long long payloadChecksum(const char *p, size_t bytes)
{
    unsigned long long h = 0x9E3779B97F4A7C15ull;
    size_t i = 0;
    for (; i + 8 <= bytes; i += 8) {
        unsigned long long w;
        std::memcpy(&w, p + i, 8);
        h = (h ^ w) * 0xFF51AFD7ED558CCDull;
        h ^= h >> 29;
    }
    unsigned long long tail = 0;
    if (i < bytes) std::memcpy(&tail, p + i, bytes - i);
    h = (h ^ tail ^ (unsigned long long)bytes) * 0xC4CEB9FE1A85EC53ull;
    return h ^ (h >> 32);
}

// Everything the kernels index with comes out of the file: child ranges, leaf triangle slots, the prim -> slot map, the level table.
// The checksum only catches accidental damage (it is not cryptographic: the cache directory is trusted like the scene files are), so
// every index is range-checked before the arrays reach the device (an out-of-range child or slot is a GPU fault or a hang in
// k_refit4 / k_trace, not a wrong pixel), the node count is bounded before anything is allocated, and every inner child must lie in
// the next level's range, which proves the depth the stack check relies on.
bool cachedTreeIsSane(const CacheHeader &h, const char *nodesBytes, const uint32_t *slotOfPrim)
{
    if (h.triSlots < h.nTris || h.triSlots >= (1u << 28) || h.rootLeafCount > 4u || h.levels > (uint32_t)kMaxLevels) return false;
    if (h.nNodes >= (1u << 26) || h.nNodes > h.nTris) return false; // (every 4-wide node stands for one binary inner node)
    if (h.rootLeafCount > 0 && h.rootLeafCount > h.triSlots) return false;
    if (h.levelStart[0] != 0u) return false;
    for (uint32_t l = 0; l < h.levels; ++l)
        if (h.levelStart[l + 1] < h.levelStart[l] || h.levelStart[l + 1] > h.nNodes) return false;
    if (h.levels > 0 && h.levelStart[h.levels] != h.nNodes) return false;
    for (uint32_t i = 0; i < h.nTris; ++i)
        if (slotOfPrim[i] >= h.triSlots) return false;
    uint32_t level = 0;
    for (uint32_t i = 0; i < h.nNodes; ++i) {
        while (level + 1 < h.levels && i >= h.levelStart[level + 1]) ++level;
        Node4 n;
        std::memcpy(&n, nodesBytes + (size_t)i * sizeof(Node4), sizeof(Node4));
        uint32_t meta;
        std::memcpy(&meta, &n.a.w, 4);
        const uint32_t nInner = (meta >> 24) & 7u, nValid = meta >> 27;
        if (nValid > 4u || nInner > nValid) return false;
        if (nInner > 0) {
            // inner children are nodes innerBase .. innerBase + nInner - 1 and lie in the NEXT level's index range (breadth-first
            // allocation): that proves the depth the header claims, which bounds the traversal stack (3 entries per level)
            const uint32_t base = n.c.z;
            if (level + 1 >= h.levels) return false;
            if (base < h.levelStart[level + 1] || base >= h.levelStart[level + 2] || nInner > h.levelStart[level + 2] - base) return false;
        }
        for (uint32_t j = nInner; j < nValid; ++j) { // leaf child j is the triangle ~(leafKey + j)
            const uint32_t slot = ~(n.c.w + j);
            if (slot >= h.triSlots) return false;
        }
    }
    return true;
}
} // namespace

// digest of everything the tree depends on: geometry bytes (hashed on the device), transforms, modes, strides
static int sceneKey(hr_ctx *c, unsigned long long *key)
{
    unsigned long long *dKey = nullptr;
    HIP_TRY(c, hipMalloc(&dKey, 8));
    hipError_t e = hipMemsetAsync(dKey, 0, 8, c->stream);
    unsigned long long host = 0xC0FFEE1234ull;
    auto mix = [&](const void *p, size_t bytes) {
        const unsigned char *b = (const unsigned char *)p;
        for (size_t i = 0; i < bytes; ++i) host = (host ^ b[i]) * 0x100000001B3ull; // FNV-1a over the small host-side fields
    };
    unsigned long long seed = 1;
    for (const Geom &g : c->geoms) {
        if (!g.alive || g.nTris() == 0) continue;
        mix(&g.nVerts, sizeof(g.nVerts)), mix(&g.nIdx, sizeof(g.nIdx)), mix(&g.mode, sizeof(g.mode)), mix(g.world, sizeof(g.world));
        mix(g.stride, sizeof(g.stride)), mix(g.off, sizeof(g.off)), mix(g.has, sizeof(g.has));
        if (e == hipSuccess) launchHashWords(c->stream, g.dBlock, g.blockBytes / 4, seed++, dKey);
    }
    unsigned long long dev = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&dev, dKey, 8, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(dKey);
    HIP_TRY(c, e);
    // (which tree a build produces also depends on the builder options: a file made with other ones is another scene's as far as the cache goes)
    mix(&c->tunePloc, sizeof(c->tunePloc)), mix(&c->tunePlocRadius, sizeof(c->tunePlocRadius));
    *key = host ^ (dev * 0x9E3779B97F4A7C15ull);
    return HR_OK;
}

// read the tree of this scene from the cache file; false: no usable file (the caller builds)
static bool loadTree(hr_ctx *c, unsigned long long key, uint32_t nTris, BuildResult *out)
{
    FILE *f = fopen(c->cachePath.c_str(), "rb");
    if (!f) return false;
    CacheHeader h;
    bool ok = fread(&h, sizeof(h), 1, f) == 1 && std::memcmp(h.magic, "HRBVHTR", 8) == 0 && h.version == kCacheVersion &&
              h.nodeBytes == sizeof(Node4) && h.key == key && h.nTris == nTris && h.nNodes > 0 && h.nNodes <= nTris && h.nNodes < (1u << 26) &&
              h.levels <= (uint32_t)kMaxLevels;
    std::vector<char> buf;
    BuildResult br{};
    if (ok) {
        const size_t nb = (size_t)h.nNodes * sizeof(Node4), bb = (size_t)h.nNodes * sizeof(Box6), sb = (size_t)nTris * 4;
        buf.resize(nb + bb + sb);
        ok = fread(buf.data(), 1, buf.size(), f) == buf.size() && fgetc(f) == EOF; // exactly the payload: nothing missing, nothing appended
        ok = ok && payloadChecksum(buf.data(), buf.size()) == h.payloadSum;
        ok = ok && cachedTreeIsSane(h, buf.data(), reinterpret_cast<const uint32_t *>(buf.data() + nb + bb));
        if (ok) {
            ok = hipMalloc(&br.nodes, nb) == hipSuccess && hipMalloc(&br.nodes32, (size_t)h.nNodes * sizeof(Node32)) == hipSuccess && hipMalloc(&br.leafKeys, (size_t)h.nNodes * sizeof(int)) == hipSuccess && hipMalloc(&br.nodeBox, bb) == hipSuccess && hipMalloc(&br.slotOfPrim, sb) == hipSuccess &&
                 hipMalloc(&br.tris, sizeof(Tri) * (size_t)h.triSlots) == hipSuccess;
            ok = ok && hipMemcpy(br.nodes, buf.data(), nb, hipMemcpyHostToDevice) == hipSuccess &&
                 hipMemcpy(br.nodeBox, buf.data() + nb, bb, hipMemcpyHostToDevice) == hipSuccess &&
                 hipMemcpy(br.slotOfPrim, buf.data() + nb + bb, sb, hipMemcpyHostToDevice) == hipSuccess &&
                 hipMemset(br.tris, 0xFF, sizeof(Tri) * (size_t)h.triSlots) == hipSuccess;
        }
    }
    fclose(f);
    if (!ok) {
        hipFree(br.nodes), hipFree(br.nodes32), hipFree(br.leafKeys), hipFree(br.nodeBox), hipFree(br.slotOfPrim), hipFree(br.tris);
        return false;
    }
    br.nNodes = (int32_t)h.nNodes, br.levels = (int32_t)h.levels, br.rootLeafCount = (int32_t)h.rootLeafCount, br.triSlots = h.triSlots;
    br.builder = h.builder == 1u ? 1 : 0, br.costRadix = h.costRadix, br.costPloc = h.costPloc;
    std::memcpy(br.levelStart, h.levelStart, sizeof(br.levelStart));
    *out = br;
    return true;
}

static void saveTree(hr_ctx *c, unsigned long long key, uint32_t nTris, const BuildResult &br)
{
    if (br.nNodes <= 0) return;
    const size_t nb = (size_t)br.nNodes * sizeof(Node4), bb = (size_t)br.nNodes * sizeof(Box6), sb = (size_t)nTris * 4;
    std::vector<char> buf(nb + bb + sb);
    if (hipMemcpy(buf.data(), br.nodes, nb, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(buf.data() + nb, br.nodeBox, bb, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(buf.data() + nb + bb, br.slotOfPrim, sb, hipMemcpyDeviceToHost) != hipSuccess)
        return;
    CacheHeader h{};
    std::memcpy(h.magic, "HRBVHTR", 8);
    h.version = kCacheVersion, h.nodeBytes = sizeof(Node4), h.key = key, h.nTris = nTris, h.nNodes = (uint32_t)br.nNodes, h.levels = (uint32_t)br.levels;
    h.rootLeafCount = (uint32_t)br.rootLeafCount, h.triSlots = br.triSlots;
    h.builder = (uint32_t)br.builder, h.costRadix = br.costRadix, h.costPloc = br.costPloc;
    std::memcpy(h.levelStart, br.levelStart, sizeof(h.levelStart));
    h.payloadSum = payloadChecksum(buf.data(), buf.size());
    const std::string tmp = c->cachePath + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return;
    const bool ok = fwrite(&h, sizeof(h), 1, f) == 1 && fwrite(buf.data(), 1, buf.size(), f) == buf.size();
    fclose(f);
    if (ok)
        rename(tmp.c_str(), c->cachePath.c_str());
    else
        remove(tmp.c_str());
}

// Device temporaries and timing events of one commit: released on every exit path.
namespace {
struct CommitScratch {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    BuildResult br{};
    bool keepBuild = false;
    ~CommitScratch()
    {
        if (e0) hipEventDestroy(e0);
        if (e1) hipEventDestroy(e1);
        if (!keepBuild) hipFree(br.nodes), hipFree(br.nodes32), hipFree(br.leafKeys), hipFree(br.tris), hipFree(br.nodeBox), hipFree(br.slotOfPrim);
    }
};
} // namespace

int hr_scene_commit(hr_ctx *c)
{
    ENTER(c);
    QUIESCE(c);
    // until this call succeeds there is no scene to render: a failed re-commit must not leave `committed` set over stale arrays
    c->committed = false, c->sceneDirty = true;
    CommitScratch cs;
    HIP_TRY(c, hipEventCreate(&cs.e0));
    HIP_TRY(c, hipEventCreate(&cs.e1));
    HIP_TRY(c, hipEventRecord(cs.e0, c->stream));
    if (!c->dConsts) {
        HIP_TRY(c, hipMalloc(&c->dConsts, sizeof(SceneConsts)));
        HIP_TRY(c, hipHostMalloc((void **)&c->hConsts, sizeof(SceneConsts), hipHostMallocDefault));
    }
    // ---- descriptors of the live geometries (their data is on the device already: hr_geom_add)
    std::vector<GeomDev> gd;
    uint32_t nTris = 0;
    bool anyExt = false;
    for (const Geom &g : c->geoms) {
        if (!g.alive || g.nTris() == 0) continue;
        GeomDev d{};
        const float *at[6];
        for (int a = 0; a < 6; ++a) at[a] = g.has[a] ? reinterpret_cast<const float *>(g.dBlock + g.off[a]) : nullptr;
        d.pos = at[0], d.nrm = at[1], d.uv = at[2], d.tan = at[3], d.bit = at[4], d.col = at[5];
        d.posStride = g.stride[0], d.nrmStride = g.stride[1], d.uvStride = g.stride[2], d.tanStride = g.stride[3], d.bitStride = g.stride[4],
        d.colStride = g.stride[5];
        d.idx = reinterpret_cast<const uint32_t *>(g.dBlock + g.off[6]);
        d.triOffset = nTris, d.nTris = g.nTris(), d.strip = g.mode == HR_TRIANGLE_STRIP;
        d.flags = (g.frontFaceCW ? TF_FRONT_CW : 0u) | (g.isOccluder ? 0u : TF_NON_OCCLUDER) | (g.has[2] ? TF_HAS_UV : 0u) |
                  ((g.has[3] && g.has[4]) ? TF_HAS_TANGENTS : 0u) | (g.has[5] ? TF_HAS_COLORS : 0u);
        d.material = (uint32_t)g.material;
        std::memcpy(d.world, g.world, sizeof(d.world));
        if (d.flags & (TF_HAS_TANGENTS | TF_HAS_COLORS)) anyExt = true;
        nTris += d.nTris;
        gd.push_back(d);
    }
    std::memset(&c->info, 0, sizeof(c->info));
    c->hScene.nodes = nullptr, c->hScene.nodes32 = nullptr, c->hScene.leafKeys = nullptr, c->hScene.tris = nullptr, c->hScene.attrs = nullptr, c->hScene.attrsExt = nullptr;
    c->hScene.nTris = 0, c->hScene.nNodes = 0, c->hScene.rootLeafCount = 0, c->hScene.rayEps = 0.0f, c->hScene.hitPad = 0.0f;
    if (nTris == 0) {
        freeTree(c);
    } else {
        int rc = ensureCap(c, &c->dG, &c->dGCap, gd.size());
        if (rc == HR_OK) rc = ensureCap(c, &c->attrs, &c->attrsCap, (size_t)nTris);
        if (rc == HR_OK && anyExt) rc = ensureCap(c, &c->attrsExt, &c->attrsExtCap, (size_t)nTris);
        if (rc != HR_OK) return rc;
        TriAttrExt *ext = anyExt ? c->attrsExt : nullptr;
        HIP_TRY(c, hipMemcpyAsync(c->dG, gd.data(), gd.size() * sizeof(GeomDev), hipMemcpyHostToDevice, c->stream));
        // A commit after transform edits only keeps the tree's topology: triangles are re-assembled straight into their leaf
        // slots and every level is refitted bottom-up.  No allocation, no host round trip before the last kernel.
        bool cacheHit = false;
        bool refit = c->tuneRefit && !c->topologyDirty && c->tree.nodes && c->treeTris == nTris && c->tree.rootLeafCount == 0;
        if (refit) {
            launchAssemble(c->stream, c->dG, (int)gd.size(), nTris, c->tree.tris, c->tree.slotOfPrim, c->attrs, ext, c->dScratch);
            launchSceneConsts(c->stream, c->dScratch, c->dConsts, nullptr);
            refitLBVH(c->stream, c->tree, nTris, c->dConsts);
            encodeNodes32(c->stream, c->tree, c->dConsts, nullptr); // (k_trace's copy of the nodes: every frame is re-encoded, the grid moves with the bounds)
            launchTriAreaSum(c->stream, c->tree.tris, c->tree.triSlots, c->dConsts);
            HIP_TRY(c, hipMemcpyAsync(c->hConsts, c->dConsts, sizeof(SceneConsts), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            // A refitted tree is only as good as its topology still fits the geometry: when the boxes have grown (an object moved
            // through or away from its neighbours; a rotation inflates axis-aligned boxes), rebuild.  Measured on an instanced scene
            // (16 objects, one travelling through the others, tools/r3_instanced_refit.py, profiles/r3j_instanced_refit.txt): the
            // refitted tree is 1.5 % slower than a fresh build at 1.11 x the built tree's box area, 4-8 % at 1.2-1.3 x, 8 % at 1.5 x,
            // 8-9 % when the object is flung away.  Box area is taken relative to the triangles' own area, which rigid motion leaves
            // alone and a scaling of the whole scene scales alike.  (Round 2 compared area / diagonal^2 with a threshold of 4: a flung
            // object grows the diagonal too, so that guard never fired.)  A rebuild of 1 M triangles costs 4.7 ms, a refit 0.3 ms.
            const SceneConsts &k = *c->hConsts;
            const float nowQ = k.triAreaSum > 0.0f ? k.areaSum / k.triAreaSum : 0.0f;
            if (c->builtAreaSum > 0.0f && nowQ > 0.01f * (float)c->tuneGuardPct * c->builtAreaSum) refit = false;
        }
        if (!refit) {
            rc = ensureCap(c, &c->trisPrim, &c->trisPrimCap, (size_t)nTris);
            if (rc != HR_OK) return rc;
            launchAssemble(c->stream, c->dG, (int)gd.size(), nTris, c->trisPrim, nullptr, c->attrs, ext, c->dScratch);
            launchSceneConsts(c->stream, c->dScratch, c->dConsts, nullptr);
            HIP_TRY(c, hipMemcpyAsync(c->hConsts, c->dConsts, sizeof(SceneConsts), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            const SceneConsts k = *c->hConsts;
            unsigned long long key = 0;
            bool fromCache = false;
            if (!c->cachePath.empty()) {
                rc = sceneKey(c, &key);
                if (rc != HR_OK) return rc;
                fromCache = loadTree(c, key, nTris, &cs.br);
            }
            if (fromCache) { // the tree is this scene's: only the triangles have to be put into their leaf slots
                launchAssemble(c->stream, c->dG, (int)gd.size(), nTris, cs.br.tris, cs.br.slotOfPrim, c->attrs, ext, c->dScratch);
                launchSceneConsts(c->stream, c->dScratch, c->dConsts, nullptr);
                cacheHit = true;
            } else {
                const BuildOptions bo{c->tunePloc, c->tunePlocRadius, (kStackLDS + kStackOvf) / 3};
                const int brc = buildLBVH(c->stream, c->trisPrim, nTris, k.lo, k.hi, k.pad, c->dConsts, &cs.br, bo);
                if (brc != 0) FAIL(c, HR_ERR_DEVICE, brc == 3 ? "LBVH refit did not reach the root" : "LBVH build failed");
                if (!c->cachePath.empty()) saveTree(c, key, nTris, cs.br);
            }
            encodeNodes32(c->stream, cs.br, c->dConsts, nullptr); // (built or read from the cache: k_trace's 32-byte copy of the nodes)
            // the traversal stack holds at most 3 entries per level of inner nodes (hr_trace.h)
            if (3 * cs.br.levels > kStackLDS + kStackOvf) FAIL(c, HR_ERR_UNSUPPORTED, "BVH deeper than the traversal stack");
            if (cs.br.triSlots >= (1u << 28)) FAIL(c, HR_ERR_UNSUPPORTED, "scene too large: triangle slots do not fit a 28-bit leaf reference");
            if ((unsigned long long)cs.br.nNodes >= (1ull << 25)) FAIL(c, HR_ERR_UNSUPPORTED, "scene too large: a 32-byte node holds its children's base index in 25 bits (2^25 nodes, ~95 M triangles)");
            freeTree(c);
            c->tree = cs.br, cs.keepBuild = true;
            c->treeTris = nTris;
            launchAreaSum(c->stream, c->tree.nodeBox, (uint32_t)c->tree.nNodes, c->dConsts);
            launchTriAreaSum(c->stream, c->tree.tris, c->tree.triSlots, c->dConsts);
            HIP_TRY(c, hipMemcpyAsync(c->hConsts, c->dConsts, sizeof(SceneConsts), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            c->builtAreaSum = c->hConsts->triAreaSum > 0.0f ? c->hConsts->areaSum / c->hConsts->triAreaSum : 0.0f;
        }
        const SceneConsts &k = *c->hConsts;
        c->nodes = c->tree.nodes, c->tris = c->tree.tris;
        c->hScene.nodes = c->nodes, c->hScene.nodes32 = c->tree.nodes32, c->hScene.leafKeys = c->tree.leafKeys, c->hScene.tris = c->tris, c->hScene.attrs = c->attrs, c->hScene.attrsExt = ext;
        gridOf(k, c->hScene.gridLo, c->hScene.gridCell, c->hScene.gridCellExp);
        c->hScene.nTris = (int)nTris, c->hScene.nNodes = c->tree.nNodes, c->hScene.rootLeafCount = c->tree.rootLeafCount;
        c->hScene.rayEps = k.eps; // 1e-4 |diagonal|, SURVEY §8a a6
        c->hScene.hitPad = 0.5f * k.pad; // (hr_trace.h: hitInTriBox)
        for (int q = 0; q < 3; ++q) c->info.aabb_min[q] = k.lo[q], c->info.aabb_max[q] = k.hi[q];
        c->info.n_triangles = nTris, c->info.n_nodes = (uint64_t)c->tree.nNodes, c->info.ray_epsilon = k.eps;
        c->info.bvh_levels = (uint32_t)c->tree.levels;
        c->info.refitted = refit ? 1u : (cacheHit ? 2u : 0u);
        c->info.box_area_ratio = (c->builtAreaSum > 0.0f && k.triAreaSum > 0.0f) ? (k.areaSum / k.triAreaSum) / c->builtAreaSum : 0.0f;
        c->info.builder = (uint32_t)c->tree.builder, c->info.cost_radix = c->tree.costRadix, c->info.cost_ploc = c->tree.costPloc;
    }
    HIP_TRY(c, hipEventRecord(cs.e1, c->stream));
    HIP_TRY(c, hipEventSynchronize(cs.e1));
    hipEventElapsedTime(&c->info.build_ms, cs.e0, cs.e1);
    c->committed = true;
    std::memset(c->stageSeen, 0, sizeof(c->stageSeen)); // (memory budget: another scene, other queue lengths)
    c->probeCountdown = 0; // (packet selector: another tree)
    c->sceneDirty = true;
    c->texDensityStale = true;
    c->topologyDirty = false, c->transformDirty = false;
    return HR_OK;
}

int hr_scene_cache(hr_ctx *c, const char *path)
{
    ENTER(c);
    c->cachePath = path ? path : "";
    return HR_OK;
}

int hr_scene_get_info(hr_ctx *c, hr_scene_info *out)
{
    ENTER(c);
    if (!c->committed || !out) FAIL(c, HR_ERR_INVALID, "scene not committed");
    *out = c->info;
    return HR_OK;
}

// --------------------------------------------------------------------------------------- textures
int hr_texture_create(hr_ctx *c, const hr_texture_desc *d, const void *pixels, hr_tex_id *out)
{
    ENTER(c);
    if (!d || !pixels || d->width <= 0 || d->height <= 0 || (d->channels != 1 && d->channels != 3 && d->channels != 4))
        FAIL(c, HR_ERR_INVALID, "bad texture descriptor");
    const size_t n = (size_t)d->width * d->height * d->channels;
    if (d->dtype != HR_TEX_U8 && d->dtype != HR_TEX_F32) FAIL(c, HR_ERR_INVALID, "bad texture dtype");
    // 8-bit data stays 8-bit in HBM (a quarter of the footprint and of the bytes per texel fetched; the sampler normalises
    // float(byte) / 255.0f on fetch, the conversion the reference's loader would otherwise leave to the RL texture unit)
    const size_t bytes = n * (d->dtype == HR_TEX_U8 ? 1 : sizeof(float));
    Texture t;
    HIP_TRY(c, hipMalloc(&t.dpx, bytes));
    HIP_TRY(c, hipMemcpy(t.dpx, pixels, bytes, hipMemcpyHostToDevice));
    t.desc = TexDesc{t.dpx, d->width, d->height, d->channels, d->wrap_s, d->wrap_t, d->filter, d->dtype, 0, nullptr, 0.0f, 0};
    t.alive = true;
    c->textures.push_back(t);
    c->sceneDirty = true;
    if (out) *out = (hr_tex_id)c->textures.size() - 1;
    return HR_OK;
}

int hr_texture_destroy(hr_ctx *c, hr_tex_id id)
{
    ENTER(c);
    if (id < 0 || id >= (int)c->textures.size() || !c->textures[id].alive) FAIL(c, HR_ERR_INVALID, "bad texture id");
    QUIESCE(c);
    hipFree(c->textures[id].dpx), hipFree(c->textures[id].dmips);
    c->textures[id] = Texture();
    if (c->envTex == id) c->envTex = -2, c->envW = c->envH = 0;
    c->sceneDirty = true;
    return HR_OK;
}

int hr_material_set(hr_ctx *c, int32_t id, const hr_material *m)
{
    ENTER(c);
    if (id < 0 || id > (1 << 20) || !m) FAIL(c, HR_ERR_INVALID, "bad material id");
    if ((int)c->materials.size() <= id) {
        hr_material none{};
        none.type = -1;
        c->materials.resize(id + 1, none);
    }
    c->materials[id] = *m;
    c->sceneDirty = true;
    return HR_OK;
}

int hr_lights_set(hr_ctx *c, const hr_lights *l)
{
    ENTER(c);
    if (!l || l->n_directional < 0 || l->n_directional > HR_MAX_DIRECTIONAL_LIGHTS || l->n_point < 0 || l->n_point > HR_MAX_POINT_LIGHTS ||
        l->n_spot < 0 || l->n_spot > HR_MAX_SPOT_LIGHTS)
        FAIL(c, HR_ERR_INVALID, "bad light block");
    c->lights = *l;
    c->sceneDirty = true;
    return HR_OK;
}

int hr_interactive_blocks_set(hr_ctx *c, const int32_t *coords, int32_t nx, int32_t ny)
{
    ENTER(c);
    if (!coords) {
        c->blockNx = c->blockNy = 0;
    } else {
        if (nx <= 0 || ny <= 0 || nx * ny > 16) FAIL(c, HR_ERR_INVALID, "block table: nx*ny must be 1..16");
        for (int i = 0; i < nx * ny; ++i)
            if (coords[2 * i] < 0 || coords[2 * i + 1] < 0) FAIL(c, HR_ERR_INVALID, "block table: negative coordinate");
        c->blockNx = nx, c->blockNy = ny;
        std::memcpy(c->blockCoords, coords, sizeof(int32_t) * 2 * (size_t)(nx * ny));
    }
    c->sceneDirty = true;
    return HR_OK;
}

// ----------------------------------------------------------------------------------- sample tables
static int setTable(hr_ctx *c, float2 **dst, const float *src, size_t n)
{
    QUIESCE(c);
    hipFree(*dst);
    *dst = nullptr;
    HIP_TRY(c, hipMalloc(dst, n * sizeof(float2)));
    if (src) HIP_TRY(c, hipMemcpy(*dst, src, n * sizeof(float2), hipMemcpyHostToDevice));
    return HR_OK;
}

int hr_sequences_set(hr_ctx *c, const float *seq, const float *ap, int32_t nSeq, int32_t len)
{
    ENTER(c);
    if (!seq || !ap || nSeq <= 0 || nSeq > 255 || len <= 0) FAIL(c, HR_ERR_INVALID, "bad sequence table");
    int rc = setTable(c, &c->dSeq, seq, (size_t)nSeq * len);
    if (rc) return rc;
    rc = setTable(c, &c->dAperture, ap, (size_t)nSeq * len);
    if (rc) return rc;
    c->nSeq = nSeq, c->seqLen = len;
    c->sceneDirty = true;
    return HR_OK;
}

int hr_seq_offsets_set(hr_ctx *c, const float *off, int32_t n)
{
    ENTER(c);
    if (!off || n <= 0) FAIL(c, HR_ERR_INVALID, "bad offsets table");
    int rc = setTable(c, &c->dSeqOffsets, off, (size_t)n);
    if (rc) return rc;
    c->nSeqOffsets = n;
    c->sceneDirty = true;
    return HR_OK;
}

// radialSobol's disk mapping (Random.h:272-287).  The reference evaluates it with the C library's sqrtf / cosf / sinf, whose
// last bit is libm-specific; it is therefore done on the host, with the same library the application itself would use, on the
// device-generated Sobol points (which are bit-exact): the aperture tables then equal the reference's bit for bit.
// (16 x maxRenderPasses points at initialisation time.)
static void radialOnHost(float2 *p, size_t count)
{
    const float two_pi = 6.28318530717958647692f;
    for (size_t i = 0; i < count; ++i) {
        const float s = p[i].x, t = p[i].y;
        const float sqrt_t = sqrtf(t);
        const float two_pi_s = two_pi * s;
        float x = sqrt_t * cosf(two_pi_s);
        float y = sqrt_t * sinf(two_pi_s);
        x = (x + 1.0f) * 0.5f;
        y = (y + 1.0f) * 0.5f;
        p[i] = make_float2(x, y);
    }
}

// the fan polygon of randomPolygonal (Random.h:299-306): `edges` vertices on the unit circle, by the platform's cosf / sinf like the
// reference's own call (same reasoning as radialOnHost: 5 to 8 values, once per table)
static void polygonOnHost(uint32_t edges, float *vx, float *vy)
{
    const float two_pi = 6.28318530717958647692f;
    const float stepSize = two_pi / (float)edges;
    for (uint32_t i = 0; i < edges; ++i) {
        const float theta = stepSize * (float)i;
        vx[i] = cosf(theta), vy[i] = sinf(theta);
    }
}
static uint32_t bokehEdges(int32_t bokeh) { return bokeh == HR_BOKEH_PENTAGON ? 5u : bokeh == HR_BOKEH_HEXAGON ? 6u : bokeh == HR_BOKEH_OCTAGON ? 8u : 0u; }

// nSeq tables of `count` sample points (sequences seq0 .. seq0 + nSeq - 1) into d + s * stride, every mode of PassGenerator.h:100-106 on the
// device: the closed-form sequences by index (k_qmc), the std:: tables and the blue-noise points by their serial algorithms (hr_tables.h)
static int sampleTablesOnDevice(hr_ctx *c, int32_t mode, uint32_t seq0, int nSeq, uint32_t count, float2 *d, size_t stride)
{
    if (mode == HR_SAMPLE_RANDOM) {
        launchMtTables(c->stream, seq0, nSeq, count, 0, nullptr, nullptr, d, stride);
    } else if (mode == HR_SAMPLE_BLUE_NOISE) {
        float2 *cand = nullptr;
        if (count > 1) HIP_TRY(c, hipMalloc(&cand, (size_t)nSeq * (count - 1u) * 30u * sizeof(float2)));
        launchBlueNoise(c->stream, (int32_t)seq0, nSeq, count, d, stride, cand);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream); // (the scratch is freed here, so the launches must be done)
        if (cand) hipFree(cand);
        HIP_TRY(c, e);
    } else {
        for (int s = 0; s < nSeq; ++s) launchQmc(c->stream, mode, seq0 + (uint32_t)s, count, d + (size_t)s * stride);
    }
    HIP_TRY(c, hipGetLastError());
    return HR_OK;
}
// the aperture tables of PassGenerator.cpp:653-676 likewise; the circular one still wants radialOnHost over the result
static int apertureTablesOnDevice(hr_ctx *c, int32_t bokeh, uint32_t seq0, int nSeq, uint32_t count, float2 *d, size_t stride)
{
    const uint32_t edges = bokehEdges(bokeh);
    if (edges == 0) return sampleTablesOnDevice(c, HR_SAMPLE_SOBOL, seq0, nSeq, count, d, stride);
    float vx[8], vy[8];
    polygonOnHost(edges, vx, vy);
    launchMtTables(c->stream, seq0, nSeq, count, edges, vx, vy, d, stride);
    HIP_TRY(c, hipGetLastError());
    return HR_OK;
}
static bool knownSampleMode(int32_t m) { return m >= HR_SAMPLE_RANDOM && m <= HR_SAMPLE_SOBOL; }
static bool knownBokeh(int32_t b) { return b >= HR_BOKEH_CIRCULAR && b <= HR_BOKEH_OCTAGON; }

static int tableToHost(hr_ctx *c, float2 *d, uint32_t count, float *out)
{
    hipError_t e = hipMemcpyAsync(out, d, (size_t)count * sizeof(float2), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(d);
    HIP_TRY(c, e);
    return HR_OK;
}

int hr_qmc_generate(hr_ctx *c, int32_t mode, uint32_t seqIndex, uint32_t count, int32_t radial, float *out)
{
    ENTER(c);
    if (!knownSampleMode(mode)) FAIL(c, HR_ERR_INVALID, "unknown sample mode");
    if (radial && mode != HR_SAMPLE_SOBOL) FAIL(c, HR_ERR_INVALID, "radial is defined for Sobol only");
    if (count == 0 || !out) FAIL(c, HR_ERR_INVALID, "bad count / output");
    float2 *d = nullptr;
    HIP_TRY(c, hipMalloc(&d, (size_t)count * sizeof(float2)));
    int rc = sampleTablesOnDevice(c, mode, seqIndex, 1, count, d, count);
    if (rc) {
        hipFree(d);
        return rc;
    }
    rc = tableToHost(c, d, count, out);
    if (rc) return rc;
    if (radial) radialOnHost(reinterpret_cast<float2 *>(out), count);
    return HR_OK;
}

int hr_aperture_generate(hr_ctx *c, int32_t bokeh, uint32_t seqIndex, uint32_t count, float *out)
{
    ENTER(c);
    if (!knownBokeh(bokeh)) FAIL(c, HR_ERR_INVALID, "unknown bokeh shape");
    if (count == 0 || !out) FAIL(c, HR_ERR_INVALID, "bad count / output");
    float2 *d = nullptr;
    HIP_TRY(c, hipMalloc(&d, (size_t)count * sizeof(float2)));
    int rc = apertureTablesOnDevice(c, bokeh, seqIndex, 1, count, d, count);
    if (rc) {
        hipFree(d);
        return rc;
    }
    rc = tableToHost(c, d, count, out);
    if (rc) return rc;
    if (bokeh == HR_BOKEH_CIRCULAR) radialOnHost(reinterpret_cast<float2 *>(out), count);
    return HR_OK;
}

int hr_sequences_generate(hr_ctx *c, int32_t sampleMode, int32_t bokeh, int32_t len)
{
    ENTER(c);
    if (!knownSampleMode(sampleMode)) FAIL(c, HR_ERR_INVALID, "unknown sample mode");
    if (!knownBokeh(bokeh)) FAIL(c, HR_ERR_INVALID, "unknown bokeh shape");
    if (len <= 0) FAIL(c, HR_ERR_INVALID, "bad sequence length");
    const int nSeq = HR_NUM_RANDOM_SEQUENCES;
    int rc = setTable(c, &c->dSeq, nullptr, (size_t)nSeq * len);
    if (rc) return rc;
    rc = setTable(c, &c->dAperture, nullptr, (size_t)nSeq * len);
    if (rc) return rc;
    rc = sampleTablesOnDevice(c, sampleMode, 0, nSeq, (uint32_t)len, c->dSeq, (size_t)len); // PassGenerator.cpp:614-637
    if (rc) return rc;
    rc = apertureTablesOnDevice(c, bokeh, 0, nSeq, (uint32_t)len, c->dAperture, (size_t)len); // :653-676
    if (rc) return rc;
    if (bokeh == HR_BOKEH_CIRCULAR) { // Sobol points from the device, disk mapping on the host (see radialOnHost)
        std::vector<float2> ap((size_t)nSeq * len);
        HIP_TRY(c, hipMemcpyAsync(ap.data(), c->dAperture, ap.size() * sizeof(float2), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        radialOnHost(ap.data(), ap.size());
        HIP_TRY(c, hipMemcpyAsync(c->dAperture, ap.data(), ap.size() * sizeof(float2), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    c->nSeq = nSeq, c->seqLen = len;
    c->sceneDirty = true;
    return HR_OK;
}

int hr_seq_offsets_generate(hr_ctx *c)
{
    ENTER(c);
    if (c->W <= 0) FAIL(c, HR_ERR_INVALID, "no frame");
    const size_t n = (size_t)c->W * c->H;
    int rc = setTable(c, &c->dSeqOffsets, nullptr, n);
    if (rc) return rc;
    launchQmc(c->stream, HR_SAMPLE_SOBOL, 0, (uint32_t)n, c->dSeqOffsets); // PassGenerator.cpp:150-159
    HIP_TRY(c, hipGetLastError());
    c->nSeqOffsets = (int)n;
    c->sceneDirty = true;
    return HR_OK;
}

int hr_multiscatter_lut_generate(hr_ctx *c, float *out, hr_tex_id *outTex)
{
    ENTER(c);
    float2 *seq = nullptr;
    float *lut = nullptr;
    HIP_TRY(c, hipMalloc(&seq, 4096 * sizeof(float2)));
    HIP_TRY(c, hipMalloc(&lut, 128 * 128 * sizeof(float)));
    launchQmc(c->stream, HR_SAMPLE_SOBOL, 0, 4096, seq); // MultiScatterUtil.cpp:102-104
    launchMultiscatterLUT(c->stream, seq, lut);
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess && out) e = hipMemcpy(out, lut, 128 * 128 * sizeof(float), hipMemcpyDeviceToHost);
    hipFree(seq);
    if (e != hipSuccess) {
        hipFree(lut);
        HIP_TRY(c, e);
    }
    if (outTex) { // loadMultiscatterTexture: LINEAR + CLAMP_TO_EDGE sampler (TextureLoader.cpp:36-41)
        Texture t;
        t.dpx = lut;
        t.desc = TexDesc{lut, 128, 128, 1, HR_WRAP_CLAMP_TO_EDGE, HR_WRAP_CLAMP_TO_EDGE, HR_FILTER_LINEAR, HR_TEX_F32, 0, nullptr, 0.0f, 0};
        t.alive = true;
        c->textures.push_back(t);
        c->sceneDirty = true;
        *outTex = (hr_tex_id)c->textures.size() - 1;
    } else {
        hipFree(lut);
    }
    return HR_OK;
}

// ------------------------------------------------------------------------------------------ pass
static int uploadScene(hr_ctx *c)
{
    if (!c->sceneDirty) return HR_OK;
    QUIESCE(c);
    if (c->dMaterialsCap < c->materials.size() || !c->dMaterials) {
        hipFree(c->dMaterials);
        c->dMaterialsCap = c->materials.size() + 16;
        HIP_TRY(c, hipMalloc(&c->dMaterials, c->dMaterialsCap * sizeof(hr_material)));
    }
    if (!c->materials.empty())
        HIP_TRY(c, hipMemcpy(c->dMaterials, c->materials.data(), c->materials.size() * sizeof(hr_material), hipMemcpyHostToDevice));
    if (c->dTexturesCap < c->textures.size() || !c->dTextures) {
        hipFree(c->dTextures);
        c->dTexturesCap = c->textures.size() + 16;
        HIP_TRY(c, hipMalloc(&c->dTextures, c->dTexturesCap * sizeof(TexDesc)));
    }
    std::vector<TexDesc> td(c->textures.size());
    for (size_t i = 0; i < td.size(); ++i) {
        td[i] = c->textures[i].desc;
        if (!c->textures[i].alive) td[i].px = nullptr;
    }
    if (!td.empty()) {
        HIP_TRY(c, hipMemcpy(c->dTextures, td.data(), td.size() * sizeof(TexDesc), hipMemcpyHostToDevice));
        launchTexLodScale(c->stream, c->dTextures, (int)td.size()); // TexDesc::lodScale, in the device's (= the oracle's) arithmetic
    }
    SceneDev &s = c->hScene;
    s.materials = c->dMaterials, s.nMaterials = (int)c->materials.size();
    s.textures = c->dTextures, s.nTextures = (int)c->textures.size();
    s.lights = c->lights;
    s.seq = c->dSeq, s.aperture = c->dAperture, s.seqOffsets = c->dSeqOffsets;
    s.nSeq = c->nSeq, s.seqLen = c->seqLen, s.nSeqOffsets = c->nSeqOffsets;
    s.envRowCdf = c->dEnvRowCdf, s.envColCdf = c->dEnvColCdf, s.envProb = c->dEnvProb;
    s.envRowGuide = c->dEnvRowGuide, s.envColGuide = c->dEnvColGuide;
    s.envW = c->envW, s.envH = c->envH, s.envMeanLum = c->envMeanLum;
    s.blockNx = c->blockNx, s.blockNy = c->blockNy;
    std::memcpy(s.blockCoords, c->blockCoords, sizeof(s.blockCoords));
    s.texDensity = c->texDensityStale ? nullptr : c->dTexDensity;
    HIP_TRY(c, hipMemcpy(c->dScene, &s, sizeof(SceneDev), hipMemcpyHostToDevice));
    // rays can outlive maxRayDepth only by passing through single-sided / alpha-masked surfaces
    c->hasPassthrough = false, c->hasGlass = false;
    for (const hr_material &m : c->materials) {
        if (m.type == HR_MAT_PBR && (!(m.flags & HR_MF_DOUBLE_SIDED) || (m.flags & HR_MF_ALPHA_MASK))) c->hasPassthrough = true;
        if (m.type == HR_MAT_GLASS) c->hasGlass = true;
    }
    c->sceneDirty = false;
    return HR_OK;
}

// HR_ESTIMATOR_ENV_MIS: (re)build the importance table of the current environment map on the device
static int ensureEnvTable(hr_ctx *c)
{
    const int id = c->lights.env_texture;
    const bool have = c->lights.env_enabled && id >= 0 && id < (int)c->textures.size() && c->textures[id].alive;
    if (!have) {
        if (c->envW != 0) {
            QUIESCE(c);
            c->envW = c->envH = 0, c->envTex = -2;
            c->sceneDirty = true;
        }
        return HR_OK;
    }
    const TexDesc &t = c->textures[id].desc;
    if (c->envTex == id && c->envW == t.w && c->envH == t.h) return HR_OK;
    if (t.w > 65535 || t.h > 65535) FAIL(c, HR_ERR_UNSUPPORTED, "environment map too large for the importance table (65535 texels per side)");
    QUIESCE(c);
    hipFree(c->dEnvRowCdf), hipFree(c->dEnvColCdf), hipFree(c->dEnvProb), hipFree(c->dEnvRowGuide), hipFree(c->dEnvColGuide);
    c->dEnvRowCdf = c->dEnvColCdf = c->dEnvProb = nullptr, c->envW = c->envH = 0, c->envTex = -2;
    c->dEnvRowGuide = c->dEnvColGuide = nullptr;
    const size_t n = (size_t)t.w * t.h;
    float *lum = nullptr, *dil = nullptr;
    uint32_t *wq = nullptr, *maxBits = nullptr;
    unsigned long long *rowSum = nullptr;
    hipError_t e = hipMalloc(&c->dEnvRowCdf, sizeof(float) * ((size_t)t.h + 1));
    if (e == hipSuccess) e = hipMalloc(&c->dEnvColCdf, sizeof(float) * (size_t)t.h * ((size_t)t.w + 1));
    if (e == hipSuccess) e = hipMalloc(&c->dEnvProb, sizeof(float) * n);
    if (e == hipSuccess) e = hipMalloc(&c->dEnvRowGuide, sizeof(uint16_t) * (kEnvRowGuide + 1));
    if (e == hipSuccess) e = hipMalloc(&c->dEnvColGuide, sizeof(uint16_t) * (size_t)t.h * (kEnvColGuide + 1));
    if (e == hipSuccess) e = hipMalloc(&lum, sizeof(float) * n);
    if (e == hipSuccess) e = hipMalloc(&dil, sizeof(float) * n);
    if (e == hipSuccess) e = hipMalloc(&wq, sizeof(uint32_t) * n);
    if (e == hipSuccess) e = hipMalloc(&rowSum, sizeof(unsigned long long) * ((size_t)t.h + 1));
    if (e == hipSuccess) e = hipMalloc(&maxBits, 16);
    if (e == hipSuccess) {
        float *dMean = reinterpret_cast<float *>(maxBits) + 1;
        launchEnvTable(c->stream, t, lum, dil, wq, rowSum, rowSum + t.h, maxBits, c->dEnvRowCdf, c->dEnvColCdf, c->dEnvProb, dMean);
        launchEnvGuides(c->stream, c->dEnvRowCdf, c->dEnvColCdf, t.w, t.h, c->dEnvRowGuide, c->dEnvColGuide);
        e = hipMemcpyAsync(&c->envMeanLum, dMean, sizeof(float), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
    hipFree(lum), hipFree(dil), hipFree(wq), hipFree(rowSum), hipFree(maxBits);
    HIP_TRY(c, e);
    c->envW = t.w, c->envH = t.h, c->envTex = id;
    c->sceneDirty = true;
    return HR_OK;
}

// HR_TEXTURE_LOD_CONE: build what the mode needs and is missing — the mip chains of the textures and the per-triangle level offset
static int ensureTextureLod(hr_ctx *c)
{
    bool quiesced = false;
    for (Texture &t : c->textures) {
        if (!t.alive || t.desc.nLevels != 0) continue;
        if (!quiesced) {
            QUIESCE(c);
            quiesced = true;
        }
        TexDesc &d = t.desc;
        int levels = 1;
        size_t elems = 0;
        for (int w = d.w, h = d.h; (w > 1 || h > 1) && d.filter != HR_FILTER_NEAREST; ++levels) {
            w = w / 2 < 1 ? 1 : w / 2, h = h / 2 < 1 ? 1 : h / 2;
            elems += (size_t)w * h * d.c;
        }
        if (levels > 1) {
            HIP_TRY(c, hipMalloc(&t.dmips, elems * sizeof(float)));
            launchMipChain(c->stream, d, levels, t.dmips);
        }
        d.nLevels = levels, d.mips = t.dmips;
        c->sceneDirty = true;
    }
    if (c->texDensityStale && c->tree.tris) {
        if (!quiesced) {
            QUIESCE(c);
            quiesced = true;
        }
        const size_t nTris = c->treeTris;
        if (c->texDensityCap < nTris || !c->dTexDensity) {
            hipFree(c->dTexDensity);
            c->dTexDensity = nullptr, c->texDensityCap = 0;
            HIP_TRY(c, hipMalloc(&c->dTexDensity, sizeof(float) * (nTris ? nTris : 1)));
            c->texDensityCap = nTris;
        }
        const uint32_t slots = c->tree.triSlots ? c->tree.triSlots : (uint32_t)nTris;
        launchTexDensity(c->stream, c->tree.tris, slots, c->attrs, c->dTexDensity);
        c->texDensityStale = false;
        c->sceneDirty = true;
    }
    if (quiesced) HIP_TRY(c, hipStreamSynchronize(c->stream));
    return HR_OK;
}

