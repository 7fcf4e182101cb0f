// hr_pipeline.inl — a section of hr_core.hip (included there, inside its extern "C" block): the pass pipeline — pass slots and ray memory,
// the macro step (tables, launches, the packet selector), batching, hr_render_pass / hr_flush, statistics of the pipeline itself.
int hr_clear(hr_ctx *c)
{
    ENTER(c);
    if (c->W <= 0) FAIL(c, HR_ERR_INVALID, "no frame");
    int rc = drainPipeline(c);
    if (rc) return rc;
    if (c->hOverflow && c->hOverflow[0]) { // a dropped-rays report: the frame starts afresh and so does the report, once nothing that could repeat it is running
        QUIESCE(c);
        c->hOverflow[0] = c->hOverflow[1] = c->hOverflow[2] = c->hOverflow[3] = 0u;
    }
    HIP_TRY(c, hipMemsetAsync(c->fb(), 0, (size_t)c->W * c->H * 4 * sizeof(float), c->stream));
    HIP_TRY(c, hipMemsetAsync(c->dStats, 0, sizeof(Stats) * kStatSlots, c->stream));
    c->resolvedAtClear = c->nextResolveOrder;
    c->snapshotEpoch++;
    if (getenv("HR_DEBUG_PIPE")) fprintf(stderr, "hr_clear %p: ray-memory growths so far %llu, waits %llu (%.2f ms)\n", (void *)c, c->dbgGrowths, c->dbgWaits, (double)c->dbgWaitNs * 1e-6);
    c->drainTimes();
    for (int k = 0; k < HR_KERNEL_COUNT; ++k) c->kernelMs[k] = 0.0f, c->kernelLaunches[k] = 0;
    return HR_OK;
}

static int allocSlot(hr_ctx *c, hr_ctx::PassSlot &ps)
{
    const size_t fbBytes = (size_t)c->W * c->H * 4 * sizeof(float);
    // (with HR_ESTIMATOR_ALL_LIGHTS the sample's further partial sums lie right behind the first: k_trace indexes one buffer)
    const hipError_t e = hipMalloc(&ps.passbuf, fbBytes * (c->allLightsUsed ? 4 : 1));
    if (e != hipSuccess) { // say what ran out: a pass slot is the unit the pipeline's memory grows in
        size_t freeB = 0, totalB = 0;
        hipMemGetInfo(&freeB, &totalB);
        c->err = "pass slot " + std::to_string(c->nSlotsAllocated + 1) + " (" + std::to_string(fbBytes >> 20) + " MiB pass buffer at " + std::to_string(c->W) + "x" +
                 std::to_string(c->H) + "): " + hipGetErrorString(e) + "; " + std::to_string(freeB >> 20) + " MiB of device memory free";
        return HR_ERR_DEVICE;
    }
    if (c->allLightsUsed) ps.passbufB = ps.passbuf + (size_t)c->W * c->H * 4;
    ps.ctr = c->dCounters + (&ps - c->slots);
    HIP_TRY(c, hipEventCreateWithFlags(&ps.evFinal, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&ps.evResolved, hipEventDisableTiming));
    ps.allocated = true;
    c->nSlotsAllocated++;
    return HR_OK;
}

// ---- the groups' ray memory (hr_ctx::Group::arena / scratch)
static size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }
static size_t rayQueueBytes(size_t cap) { return 4 * align256(cap * 16); }
static size_t shadowQueueBytes(size_t cap) { return 3 * align256(cap * 16); }
static RayQueue carveRayQueue(char *&p, size_t cap)
{
    RayQueue q;
    const size_t n = align256(cap * 16);
    q.A = (float4 *)p, q.B = (float4 *)(p + n), q.C = (float4 *)(p + 2 * n), q.D = (int4 *)(p + 3 * n);
    p += 4 * n;
    return q;
}
static ShadowQueue carveShadowQueue(char *&p, size_t cap)
{
    ShadowQueue q;
    const size_t n = align256(cap * 16);
    q.A = (float4 *)p, q.B = (float4 *)(p + n), q.C = (float4 *)(p + 2 * n);
    p += 3 * n;
    return q;
}
// a region that is too small is replaced once everything the group has enqueued is done (what it held is dead by then: a step's
// scratch dies with the step, and arena[t & 1] holds the rays step t - 2 emitted, which step t - 1 consumed)
static int ensureRegion(hr_ctx *c, hr_ctx::Group &G, hr_ctx::Group::Region &r, size_t need, const char *what)
{
    if (need <= r.cap) return HR_OK;
    c->dbgGrowths++, c->dbgGrowBytes += need;
    if (getenv("HR_DEBUG_PIPE")) fprintf(stderr, "  grow %s: need %.1f MiB, had %.1f MiB (step %llu)\n", what, (double)need / 1048576.0, (double)r.cap / 1048576.0, G.stepCounter);
    HIP_TRY(c, hipStreamSynchronize(G.stream));
    const size_t hadCap = r.cap;
    hipFree(r.base);
    r.base = nullptr, r.cap = 0;
    // a third of headroom: counts vary from pass to pass, and while the pipeline fills (the first depth + 2 steps of a render) every
    // step carries one more generation of passes — for the benchmark soup the steady state needs 27 % more than the step that
    // triggered the last growth (profiles/r4m_mem.txt); a step that needs more regrows once more
    size_t want = need + need / 3;
    if (hadCap && !c->memBudget && want < hadCap + hadCap / 2) want = hadCap + hadCap / 2; // (a region that has to grow again grows by half at least: few events; under a memory budget only by what is needed)
    want = (want + ((size_t)2 << 20)) & ~(((size_t)2 << 20) - 1);
    hipError_t e = hipMalloc((void **)&r.base, want);
    size_t got = want;
    if (e != hipSuccess) {
        (void)hipGetLastError();
        got = align256(need);
        e = hipMalloc((void **)&r.base, got);
    }
    if (e != hipSuccess) {
        size_t freeB = 0, totalB = 0;
        hipMemGetInfo(&freeB, &totalB);
        c->err = std::string("ray memory (") + what + ", " + std::to_string(need >> 20) + " MiB for one macro step at " + std::to_string(c->W) + "x" + std::to_string(c->H) +
                 "): " + hipGetErrorString(e) + "; " + std::to_string(freeB >> 20) + " MiB of device memory free";
        return HR_ERR_DEVICE;
    }
    r.cap = got;
    return HR_OK;
}

// passes that hold a slot: in flight, or finished and waiting for their turn to resolve
static int occupiedSlots(const hr_ctx *c, int group)
{
    int n = 0;
    for (const hr_ctx::PassSlot &ps : c->slots) n += ((ps.active || ps.finished) && (group < 0 || ps.group == group)) ? 1 : 0;
    return n;
}

static int slotLimit(const hr_ctx *c);
static int activePasses(const hr_ctx *c)
{
    int n = 0;
    for (const hr_ctx::PassSlot &ps : c->slots) n += ps.active ? 1 : 0;
    return n;
}

// Finished passes are added to the frame on the caller's stream, strictly in pass order (float addition order is
// part of the arithmetic contract), whatever order the groups finished them in.
static int resolveReady(hr_ctx *c)
{
    FrameDev fr = c->frame;
    fr.fb = c->fb();
    const LaunchCfg cfg = c->cfg(c->stream);
    for (;;) {
        // collect the passes whose turn it is (up to kMaxBatch) and add them with one launch
        PassBufList bufs{};
        hr_ctx::PassSlot *ready[kMaxBatch];
        while (bufs.n < kMaxBatch) {
            hr_ctx::PassSlot *next = nullptr;
            const unsigned long long want = c->nextResolveOrder + (unsigned long long)bufs.n;
            for (hr_ctx::PassSlot &ps : c->slots)
                if ((ps.active || ps.finished) && ps.order == want) next = &ps;
            if (!next || !next->finished) break;
            ready[bufs.n] = next;
            bufs.bufB[bufs.n] = next->pp.estimator == HR_ESTIMATOR_ALL_LIGHTS ? next->passbufB : nullptr;
            bufs.buf[bufs.n++] = next->passbuf;
        }
        if (bufs.n == 0) {
            // nothing requested is unfinished any more: the age of "the oldest waiting request" starts afresh with the next request
            // (stamped only in hr_render_pass, it used to survive every pass that completed the normal way, so that 4 ms after the
            // first request EVERY progressive read-back that found the streams idle drained a partly filled batch)
            if (c->pendingInject.empty() && occupiedSlots(c) == 0) c->oldestWaitingNs = 0;
            return HR_OK;
        }
        for (int k = 0; k < bufs.n; ++k) {
            bool seen = false;
            for (int j = 0; j < k; ++j) seen = seen || ready[j]->finalEv == ready[k]->finalEv;
            if (!seen) HIP_TRY(c, hipStreamWaitEvent(c->stream, ready[k]->finalEv, 0));
        }
        c->timeBegin(HR_KERNEL_RESOLVE, c->stream);
        launchResolve(cfg, fr, bufs);
        c->timeEnd(c->stream);
        HIP_TRY(c, hipEventRecord(ready[bufs.n - 1]->evResolved, c->stream));
        for (int k = 0; k < bufs.n; ++k) {
            ready[k]->resolvedEv = ready[bufs.n - 1]->evResolved;
            ready[k]->finished = false, ready[k]->everResolved = true;
            ready[k]->resolvedAt = c->nextResolveOrder;
            c->nextResolveOrder++;
        }
    }
}

// wait for the queue lengths step (want - 1) of this group reports when its k_trace starts (Group::hCounts)
static int waitCounts(hr_ctx *c, hr_ctx::Group &G, int ring, unsigned long long want)
{
    const auto t0 = std::chrono::steady_clock::now();
    c->dbgWaits++;
    for (unsigned spins = 0;; ++spins) {
        if (G.hSeq[ring] == want) {
            if (spins) c->dbgWaitSpun++, c->dbgWaitNs += (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
            break;
        }
        if ((spins & 255u) == 255u) {
            const hipError_t q = hipStreamQuery(G.stream);
            if (q == hipSuccess) { // everything enqueued has run: the report must have arrived
                if (G.hSeq[ring] == want) break;
                FAIL(c, HR_ERR_DEVICE, "internal: a step's queue lengths never arrived");
            }
            if (q != hipErrorNotReady) HIP_TRY(c, q);
            std::this_thread::yield();
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return HR_OK;
}

// One macro step of pipeline group g: (raygen of the injected passes) -> trace of every in-flight pass of the group ->
// shade; passes whose last stage this was become `finished`.
static const int kProbeEvery = 64; // injecting steps between two probes of the packet selector
static int stagesOf(const hr_ctx *c, const hr_pass_params &pp);
static int packetLog2(const hr_ctx *c);
static bool packetsInUse(const hr_ctx *c);
static int macroStep(hr_ctx *c, int g, int nInject)
{
    static const bool dbgT = getenv("HR_DEBUG_STEPTIMES") != nullptr;
    auto nowUs = [] { return (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() * 1e-3; };
    const double tA = dbgT ? nowUs() : 0.0;
    double tB = 0, tC = 0, tD = 0, tE = 0;
    hr_ctx::Group &G = c->groups[g];
    const LaunchCfg cfg = c->cfg(G.stream);
    FrameDev fr = c->frame;
    fr.fb = c->fb();
    if (G.needUserSync) { // state set up on the caller's stream (scene, tables, cleared buffers) must be visible
        HIP_TRY(c, hipEventRecord(G.evUser, c->stream));
        HIP_TRY(c, hipStreamWaitEvent(G.stream, G.evUser, 0));
        G.needUserSync = false;
    }
    int injectedSlots[kMaxSegs];
    hipEvent_t waited[kMaxSegs];
    int nInjected = 0;
    for (int k = 0; k < nInject; ++k) {
        const hr_pass_params pp = c->pendingInject.front();
        c->pendingInject.pop_front();
        // Reuse the free slot whose pass was resolved longest ago: the injection waits for that resolve, and a slot freed
        // by the step just enqueued would chain this group's step behind the other group's (no overlap).  A slot resolved
        // only recently is passed over for fresh memory while the slot budget allows.
        int slot = -1, fresh = -1;
        for (int i = 0; i < kMaxSlots; ++i) {
            const hr_ctx::PassSlot &cand = c->slots[i];
            if (cand.active || cand.finished) continue;
            if (!cand.allocated) {
                if (fresh < 0) fresh = i;
            } else if (slot < 0 || cand.resolvedAt < c->slots[slot].resolvedAt) {
                slot = i;
            }
        }
        const unsigned long long recent = 2ull * (unsigned long long)c->nGroups * (unsigned long long)(nInject > 0 ? nInject : 1);
        if (fresh >= 0 && c->nSlotsAllocated < slotLimit(c) && (slot < 0 || (c->nGroups > 1 && c->nextResolveOrder - c->slots[slot].resolvedAt < recent)))
            slot = fresh;
        if (slot < 0) slot = fresh;
        if (slot < 0) FAIL(c, HR_ERR_INVALID, "internal: no free pass slot");
        hr_ctx::PassSlot &ps = c->slots[slot];
        if (!ps.allocated) {
            int rc = allocSlot(c, ps);
            if (rc) return rc;
        }
        if (ps.everResolved) { // the pass buffer is free again once the launch that resolved it has run
            bool seen = false;
            for (int j = 0; j < nInjected; ++j) seen = seen || waited[j] == ps.resolvedEv;
            if (!seen) HIP_TRY(c, hipStreamWaitEvent(G.stream, ps.resolvedEv, 0));
        }
        waited[nInjected] = ps.everResolved ? ps.resolvedEv : nullptr;
        ps.active = true, ps.finished = false, ps.group = g, ps.step = 0, ps.nIter = pp.max_ray_depth + 1, ps.pp = pp;
        ps.qcur = RayQueue{}, ps.scur = ShadowQueue{}, ps.capCur = 0, ps.sCapCur = 0;
        ps.order = c->injected++;
        injectedSlots[nInjected++] = slot;
    }
    for (int j0 = 0; j0 < nInjected; j0 += kMaxBatch) { // the injected passes' counters back to zero, one launch
        CounterList cl{};
        for (int j = j0; j < nInjected && cl.n < kMaxBatch; ++j) cl.ctr[cl.n++] = c->slots[injectedSlots[j]].ctr;
        launchZeroCounters(cfg, cl);
    }
    // Pass-through rays (back faces of single-sided materials, alpha masks: physicallyBased.rlsl:70-108) are not bounded by
    // maxRayDepth, so in such scenes a pass runs until its closest-hit queue is empty.  The queue lengths come from the snapshot
    // taken two macro steps ago (see Group::hQCount): a slot about to run stage `st` then knows the lengths of stages <= st - 1;
    // if stage st - 1 had no rays, it emitted nothing and the pass was complete with the steps already enqueued.
    if (c->hasPassthrough) {
        const unsigned long long N = G.stepCounter;
        if (N >= 2 && G.statusUsed[(N - 2) % kTableRing]) {
            const int ring = (int)((N - 2) % kTableRing);
            HIP_TRY(c, hipEventSynchronize(G.statusEv[ring]));
            hr_ctx::PassSlot *endedEarly[kMaxSlots];
            int nEndedEarly = 0;
            const uint32_t *snap = G.hQCount + (size_t)ring * kMaxSlots * kMaxBounceSlots;
            for (int i = 0; i < kMaxSlots; ++i) {
                hr_ctx::PassSlot &ps = c->slots[i];
                if (!ps.active || ps.group != g || G.statusOrder[ring][i] != ps.order + 1ull) continue;
                const int st = ps.step;
                const bool empty = st >= 2 && snap[(size_t)i * kMaxBounceSlots + ((st - 1) % kMaxBounceSlots)] == 0u;
                if (empty) {
                    ps.active = false, ps.finished = true;
                    endedEarly[nEndedEarly++] = &ps;
                }
            }
            hr_ctx::PassSlot *owner = nullptr; // the newest of them: it is resolved last, so its event outlives the others' waits
            for (int k = 0; k < nEndedEarly; ++k)
                if (!owner || endedEarly[k]->order > owner->order) owner = endedEarly[k];
            if (owner) HIP_TRY(c, hipEventRecord(owner->evFinal, G.stream));
            for (int k = 0; k < nEndedEarly; ++k) endedEarly[k]->finalEv = owner->evFinal;
        }
    }
    if (dbgT) tB = nowUs();
    // table of the group's in-flight passes, oldest first
    int order[kMaxSlots], n = 0;
    for (int i = 0; i < kMaxSlots; ++i)
        if (c->slots[i].active && c->slots[i].group == g) order[n++] = i;
    for (int a = 1; a < n; ++a)
        for (int b = a; b > 0 && c->slots[order[b]].order < c->slots[order[b - 1]].order; --b) std::swap(order[b], order[b - 1]);
    if (n == 0) return resolveReady(c);
    if (n > kMaxSegs) FAIL(c, HR_ERR_INVALID, "internal: too many passes in one group");
    const unsigned long long stepIdx = G.stepCounter++;
    const int ring = (int)(stepIdx % kTableRing);
    if (G.tableUsed[ring]) HIP_TRY(c, hipEventSynchronize(G.tableCopied[ring])); // staging entry free again (4 steps old)
    // ---- ray memory of this step (Group::arena): every queue sized by an upper bound of what can arrive in it
    const uint32_t P = c->tuneOverflowTest == 1 ? c->queueCapacity / 8u + 1u : (c->queueCapacity ? c->queueCapacity : 1u); // (ovf=1, TEST ONLY: camera rays do not fit)
    const size_t kS = c->allLightsUsed ? 4 : 1;
    uint32_t boundIn[kMaxSegs];
    {
        bool wanted = false;
        for (int k = 0; k < n; ++k) wanted = wanted || c->slots[order[k]].step > 0;
        int idxOfSlot[kMaxSlots];
        const int prev = (int)((stepIdx + kTableRing - 1) % kTableRing);
        if (wanted && stepIdx > 0) {
            int rc = waitCounts(c, G, prev, stepIdx); // (step stepIdx - 1 wrote stepIdx: its number + 1)
            if (rc) return rc;
            if (g == 0 && c->probePending && stepIdx > c->probeStep) { // (steps from the probe's own on report the totals) complete once every wave of the probe has counted itself
                const unsigned long long pk = G.hProbe[4 * prev], ry = G.hProbe[4 * prev + 1], done = G.hProbe[4 * prev + 2], nr = G.hProbe[4 * prev + 3];
                if (done - c->probeSeen[2] >= c->probeWaves) {
                    const unsigned long long dPk = pk - c->probeSeen[0], dRy = ry - c->probeSeen[1], dNr = nr - c->probeSeen[3];
                    c->probePending = false, c->probeSeen[0] = pk, c->probeSeen[1] = ry, c->probeSeen[2] = done, c->probeSeen[3] = nr;
                    if (dRy > 0) {
                        c->lastUnion = (double)dPk / (double)dRy;
                        c->packetsOn = c->lastUnion * 100.0 < (double)c->tunePacketUnion;
                        c->lastOwnPerRay = dNr ? (double)dRy / (double)dNr : 0.0;
                    }
                    if (getenv("HR_DEBUG_PIPE")) fprintf(stderr, "packet probe of step %llu (seen at step %llu): union %.3f, %.1f child boxes entered per ray -> packets %s\n", c->probeStep, stepIdx, c->lastUnion, c->lastOwnPerRay, c->packetsOn ? "on" : "off");
                }
            }
            for (int i = 0; i < kMaxSlots; ++i) idxOfSlot[i] = -1;
            for (int j = 0; j < G.countN[prev]; ++j) idxOfSlot[G.countSlot[prev][j]] = j;
        }
        for (int k = 0; k < n; ++k) {
            const hr_ctx::PassSlot &ps = c->slots[order[k]];
            uint32_t b = P;
            if (ps.step > 0) {
                b = ps.capCur; // (what its queue can hold is a bound too: used when the pass was not in the previous step's table)
                const int j = (wanted && stepIdx > 0) ? idxOfSlot[order[k]] : -1;
                if (j >= 0 && G.countOrder[prev][j] == ps.order + 1ull) {
                    const uint32_t seen = G.hCounts[(size_t)prev * kMaxSegs + j]; // length of its closest-hit queue one stage ago
                    b = seen < b ? seen : b;
                    if (ps.step == 1 && seen < P) c->lastCameraCount = seen; // (camera rays that passed the root cull: what a packet kernel traces per pass)
                }
            }
            // TEST ONLY (HR_TUNE="ovf=": tests/test_gpu_parity.py forces every kind of overflow once): half of what the bound should be
            if (c->tuneOverflowTest == 2 && ps.step == 1) b = b / 2u + 1u;
            boundIn[k] = b;
        }
    }
    if (dbgT) tC = nowUs();
    size_t needArena = 0, needScratch = 0;
    for (int k = 0; k < n; ++k) {
        const hr_ctx::PassSlot &ps = c->slots[order[k]];
        const bool closest = c->hasPassthrough || ps.step < ps.nIter;
        if (ps.step == 0) needScratch += rayQueueBytes(P);
        if (closest) {
            needScratch += align256((size_t)boundIn[k] * hitRecordSize()) + align256((size_t)boundIn[k] * 4);
            needArena += rayQueueBytes(boundIn[k]) + shadowQueueBytes((size_t)boundIn[k] * kS);
        }
    }
    if (c->memBudget) { // what this step carves per pass and stage (budgetBytesPerPass)
        double sumA[kMaxBounceSlots] = {0}, sumS[kMaxBounceSlots] = {0};
        int cnt[kMaxBounceSlots] = {0};
        for (int k = 0; k < n; ++k) {
            const hr_ctx::PassSlot &ps = c->slots[order[k]];
            const int st = ps.step < kMaxBounceSlots ? ps.step : kMaxBounceSlots - 1;
            const bool closest = c->hasPassthrough || ps.step < ps.nIter;
            cnt[st]++;
            if (ps.step == 0) sumS[st] += (double)rayQueueBytes(P);
            if (closest) {
                sumS[st] += (double)(align256((size_t)boundIn[k] * hitRecordSize()) + align256((size_t)boundIn[k] * 4));
                sumA[st] += (double)(rayQueueBytes(boundIn[k]) + shadowQueueBytes((size_t)boundIn[k] * kS));
            }
        }
        for (int st = 0; st < kMaxBounceSlots; ++st)
            if (cnt[st]) {
                const double a = sumA[st] / cnt[st], sc = sumS[st] / cnt[st];
                c->stageArenaSeen[st] = (c->stageSeen[st] && c->stageArenaSeen[st] > a) ? c->stageArenaSeen[st] : a;
                c->stageScratchSeen[st] = (c->stageSeen[st] && c->stageScratchSeen[st] > sc) ? c->stageScratchSeen[st] : sc;
                c->stageSeen[st] = true;
            }
    }
    hr_ctx::Group::Region &arena = G.arena[stepIdx & 1ull];
    {
        G.arenaHighWater = needArena > G.arenaHighWater ? needArena : G.arenaHighWater;
        int rc = ensureRegion(c, G, arena, G.arenaHighWater, "rays emitted by a step");
        if (rc == HR_OK) rc = ensureRegion(c, G, G.scratch, needScratch, "camera rays and hit records of a step");
        if (rc) {
            // out of device memory: nothing of this step has been enqueued except the counters' reset.  The passes it was to inject go
            // back to the head of the request queue (their slots are free again), so that a later call — after the caller has released
            // memory — injects them properly instead of tracing queues no k_raygen ever filled.
            for (int j = nInjected - 1; j >= 0; --j) {
                hr_ctx::PassSlot &ps = c->slots[injectedSlots[j]];
                c->pendingInject.push_front(ps.pp);
                ps.active = false;
                c->injected--;
            }
            G.stepCounter--;
            return rc;
        }
    }
    char *pArena = arena.base, *pScratch = G.scratch.base;
    StepTable &tbl = G.hTables[ring];
    std::memset(tbl.heads, 0, sizeof(tbl.heads));
    std::memset(tbl.clkStart, 0xFF, sizeof(tbl.clkStart)), std::memset(tbl.clkEnd, 0, sizeof(tbl.clkEnd));
    tbl.headsLog2 = (uint32_t)(c->tuneHeads < 0 ? 0 : (c->tuneHeads > 6 ? 6 : c->tuneHeads));
    tbl.nSeg = n;
    tbl.refillLanes = c->tuneRefill, tbl.triPhaseLanes = c->tuneTri;
    tbl.fetchMax = c->tuneFetchMax > 0 ? c->tuneFetchMax : 1, tbl.fetchMin = c->tuneFetchMin > 0 ? c->tuneFetchMin : 1;
    tbl.staticPerWave = c->tuneStaticDeal, tbl.hasGlass = c->hasGlass ? 1 : 0;
    tbl.primaryFromSeg = n, tbl.fetchMaxPrimary = ((c->tuneFetchPrimary > 0 ? c->tuneFetchPrimary : 1) & 0xFFFF) | ((c->tuneFetchGate & 0xFFFF) << 16); // (primaryFromSeg is set below, once the injected passes' places in the table are known)
    int injectedSegs[kMaxSegs];
    int nInjectedSegs = 0;
    for (int k = 0; k < n; ++k) {
        hr_ctx::PassSlot &ps = c->slots[order[k]];
        SegDev &sg = tbl.seg[k];
        const int st = ps.step;
        const bool closest = c->hasPassthrough || st < ps.nIter;
        sg.qin = st == 0 ? carveRayQueue(pScratch, P) : ps.qcur; // (a new pass's camera rays live for this step only)
        sg.sqIn = ps.scur;                                        // (nothing to trace there in a pass's first step: sCountIn is the zero word)
        sg.qinCap = st == 0 ? P : ps.capCur, sg.sInCap = st == 0 ? 0u : ps.sCapCur, sg.sOutCap = 0u;
        sg.qout = RayQueue{}, sg.sqOut = ShadowQueue{}, sg.hits = nullptr, sg.hitIdx = nullptr;
        if (closest) {
            sg.hits = (HitRec *)pScratch, pScratch += align256((size_t)boundIn[k] * hitRecordSize());
            sg.hitIdx = (uint32_t *)pScratch, pScratch += align256((size_t)boundIn[k] * 4);
            sg.qout = carveRayQueue(pArena, boundIn[k]);
            sg.sqOut = carveShadowQueue(pArena, (size_t)boundIn[k] * kS);
            ps.qcur = sg.qout, ps.scur = sg.sqOut, ps.capCur = boundIn[k];
            sg.sOutCap = (uint32_t)((size_t)boundIn[k] * kS);
            if (c->tuneOverflowTest == 3 && st == 0) sg.sOutCap = sg.sOutCap / 8u + 1u; // TEST ONLY: the first hits' occlusion rays do not fit
            ps.sCapCur = sg.sOutCap;
        }
        sg.passbuf = ps.passbuf;
        sg.passbufB = ps.pp.estimator == HR_ESTIMATOR_ALL_LIGHTS ? ps.passbufB : nullptr;
        // The per-stage counters are a ring: a chain of pass-through rays (stacked single-sided sheets seen from behind, alpha holes:
        // physicallyBased.rlsl:70-108 re-emits without a depth bound) can outlive any fixed number of stages, so from stage
        // kMaxBounceSlots - 1 on the entries this step appends to are cleared first (their previous use lies a whole ring back).
        const int R = kMaxBounceSlots;
        if (st + 1 >= R) {
            HIP_TRY(c, hipMemsetAsync(&ps.ctr->qCount[(st + 1) % R], 0, sizeof(uint32_t), G.stream));
            if (st >= R) {
                HIP_TRY(c, hipMemsetAsync(&ps.ctr->sCount[st % R], 0, sizeof(uint32_t), G.stream));
                HIP_TRY(c, hipMemsetAsync(&ps.ctr->pCount[st % R], 0, sizeof(uint32_t), G.stream));
                HIP_TRY(c, hipMemsetAsync(&ps.ctr->gCount[st % R], 0, sizeof(uint32_t), G.stream));
            }
        }
        sg.qCountIn = &ps.ctr->qCount[st % R];
        sg.sCountIn = st > 0 ? &ps.ctr->sCount[(st - 1) % R] : c->dZero;
        sg.qCountOut = &ps.ctr->qCount[(st + 1) % R];
        sg.sCountOut = &ps.ctr->sCount[st % R];
        sg.pCount = &ps.ctr->pCount[st % R], sg.gCount = &ps.ctr->gCount[st % R];
        sg.hitCap = closest ? boundIn[k] : 0u, sg.packets = 0; // (capacity of hits, the hit list and qout: what was carved above)
        sg.pp = ps.pp;
        sg.closestEnabled = closest ? 1 : 0;
        G.countSlot[ring][k] = order[k], G.countOrder[ring][k] = ps.order + 1ull;
        for (int j = 0; j < nInjected; ++j)
            if (order[k] == injectedSlots[j]) injectedSegs[nInjectedSegs++] = k;
    }
    if (nInjectedSegs > 0) tbl.primaryFromSeg = injectedSegs[0]; // (the table is in pass order: the passes injected now are its last entries)
    // packet selector (above): do the injected passes' camera rays travel as packets (k_raygen_packets), beside k_trace or in front of it, and does this step carry a probe?
    int probeSeg = -1;
    bool packetsNow = packetsInUse(c) && nInjectedSegs > 0 && tbl.seg[injectedSegs[0]].pp.interactive_mode == 0; // (interactive sub-passes of one sample share no pixels)
    if (c->tunePackets == 2 && g == 0 && nInjectedSegs > 0 && !c->probePending && tbl.seg[injectedSegs[0]].pp.interactive_mode == 0) {
        const hr_pass_params &pp = tbl.seg[injectedSegs[0]].pp;
        float cam[21] = {pp.fov_tan, pp.aspect_ratio, pp.focus_distance, pp.aperture_radius};
        std::memcpy(cam + 4, pp.view_matrix, sizeof(pp.view_matrix));
        cam[20] = (float)pp.interactive_mode;
        // another camera sees another part of the tree: probe again, but not more often than every eighth injecting step (a camera in motion)
        if (std::memcmp(cam, c->probeCamera, sizeof(cam)) != 0 && c->probeCountdown > 0 && c->probeCountdown <= kProbeEvery - 8) c->probeCountdown = 0;
        if (c->probeCountdown <= 0) {
            probeSeg = injectedSegs[0];
            std::memcpy(c->probeCamera, cam, sizeof(cam));
        } else {
            c->probeCountdown--;
        }
    }
    const bool corunNow = packetsNow && (c->tuneCorun == 2 || (c->tuneCorun == 1 && c->lastOwnPerRay >= (double)c->tuneCorunMin));
    for (int j = 0; j < nInjectedSegs; ++j)
        if (packetsNow) tbl.seg[injectedSegs[j]].packets = corunNow ? 2 : 1;
    tbl.hostCameraCount = corunNow ? G.dCounts + (size_t)kTableRing * kMaxSegs : nullptr;
    tbl.probe = c->dProbe, tbl.hostProbe = (g == 0 && (c->probePending || probeSeg >= 0)) ? G.dProbeHost + 4 * ring : nullptr; // (reported only while a probe is awaited)
    G.countN[ring] = n;
    tbl.hostCounts = G.dCounts + (size_t)ring * kMaxSegs, tbl.hostSeq = G.dSeq + ring, tbl.seqValue = stepIdx + 1ull;
    tbl.stepLog = c->dStepLog, tbl.nInjectedNow = (uint32_t)nInjected, tbl.group = (uint32_t)g;
    tbl.hostOverflow = c->dOverflowHost;
    StepTable *dTbl = G.dTables + ring;
    const size_t tblBytes = offsetof(StepTable, seg) + (size_t)n * sizeof(SegDev);
    if (dbgT) tD = nowUs();
    if (c->tuneTableKernel)
        launchFetchTable(G.stream, G.dTablesHost + ring, dTbl, (tblBytes + 15) & ~(size_t)15);
    else
        HIP_TRY(c, hipMemcpyAsync(dTbl, &tbl, tblBytes, hipMemcpyHostToDevice, G.stream));
    if (dbgT) tE = nowUs();
    HIP_TRY(c, hipEventRecord(G.tableCopied[ring], G.stream));
    G.tableUsed[ring] = true;
    if (c->pending.size() > 8192) c->drainTimes();
    bool timing = false; // (the kernels of a step are enqueued back to back: n + 1 timing events for n kernels)
    bool forked = false;
    for (int j0 = 0; j0 < nInjectedSegs;) { // one launch for the passes injected this step
        // (as packets: ray generation and the camera rays' traversal in one launch per group of 16, 8, 4, 2, 1 passes — the bucket
        // HR_KERNEL_RAYGEN then holds both, HR_KERNEL_TRACE and the step's device clock stay k_trace's own)
        int take = 1;
        if (packetsNow)
            while (2 * take <= nInjectedSegs - j0 && 2 * take <= kMaxBatch) take *= 2;
        else
            take = nInjectedSegs - j0 < kMaxBatch ? nInjectedSegs - j0 : kMaxBatch;
        SegList segs{};
        for (int j = j0; j < j0 + take; ++j) segs.seg[segs.n++] = injectedSegs[j];
        j0 += take;
        bool uniformParams = packetsNow;
        for (int j = 1; j < segs.n && uniformParams; ++j) { // (the usual batch: one camera, one set of options, consecutive sample indices)
            hr_pass_params a = tbl.seg[segs.seg[0]].pp, b = tbl.seg[segs.seg[j]].pp;
            a.sample_index = b.sample_index = 0;
            uniformParams = std::memcmp(&a, &b, sizeof(a)) == 0;
        }
        if (corunNow) { // beside k_trace: fork after the table copy, join before the shading kernels
            LaunchCfg cb = cfg;
            cb.stream = G.streamB;
            if (!forked) {
                HIP_TRY(c, hipEventRecord(G.evFork, G.stream));
                HIP_TRY(c, hipStreamWaitEvent(G.streamB, G.evFork, 0));
                forked = true;
            }
            launchRaygenPackets(cb, c->dScene, c->nodes, c->tris, dTbl, segs, fr, c->dStats, uniformParams);
            continue;
        }
        if (timing)
            c->timeNext(HR_KERNEL_RAYGEN, G.stream);
        else
            c->timeBegin(HR_KERNEL_RAYGEN, G.stream);
        timing = true;
        if (packetsNow)
            launchRaygenPackets(cfg, c->dScene, c->nodes, c->tris, dTbl, segs, fr, c->dStats, uniformParams);
        else
            launchRaygen(cfg, c->dScene, dTbl, segs, fr, c->dStats);
    }
    if (c->tuneShadowProbe) { // measurement only: the coherence of the occlusion rays this step's k_trace is about to trace
        if (!c->dShadowProbe) {
            HIP_TRY(c, hipMalloc(&c->dShadowProbe, 64));
            HIP_TRY(c, hipMemsetAsync(c->dShadowProbe, 0, 64, G.stream));
        }
        SegList sl{};
        uint32_t most = 0;
        for (int k = 0; k < n; ++k) {
            const hr_ctx::PassSlot &ps = c->slots[order[k]];
            if (ps.step >= 1 && (c->tuneShadowProbe == 2 || ps.step == 1) && sl.n < kMaxBatch) sl.seg[sl.n++] = k, most = tbl.seg[k].sInCap > most ? tbl.seg[k].sInCap : most;
        }
        launchShadowProbe(G.stream, c->dScene, c->nodes, c->tris, dTbl, sl, most, c->dShadowProbe);
    }
    if (forked) HIP_TRY(c, hipEventRecord(G.evJoin, G.streamB));
    if (timing)
        c->timeNext(HR_KERNEL_TRACE, G.stream);
    else
        c->timeBegin(HR_KERNEL_TRACE, G.stream);
    if (probeSeg >= 0) {
        // (probeSeen holds the totals of the report the previous decision was taken on: probes never overlap, that probe was complete)
        HIP_TRY(c, hipEventRecord(c->evProbeA, G.stream));
        HIP_TRY(c, hipStreamWaitEvent(c->probeStream, c->evProbeA, 0));
        c->probeWaves = (unsigned long long)launchPacketProbe(c->probeStream, c->dScene, c->nodes, c->tris, tbl.seg[probeSeg].pp, c->tuneProbeLog2 >= 0 ? c->tuneProbeLog2 : packetLog2(c), fr, c->dProbe);
        HIP_TRY(c, hipEventRecord(c->evProbeB, c->probeStream));
        c->probeGuard = true, c->probePending = true, c->probeStep = stepIdx, c->probeCountdown = kProbeEvery;
    }
    {
        LaunchCfg ct = cfg;
        if (forked) {
            // camera rays of this step (what the passes injected before sent through the root cull, or half the pixels while unknown)
            // against the rays k_trace carries (two per entry of the closest-hit queues' bounds: the ray and its occlusion ray)
            double others = 0.0;
            for (int k = 0; k < n; ++k)
                if (c->slots[order[k]].step > 0) others += 2.0 * (double)((c->slots[order[k]].step == 1 && c->lastCameraCount && boundIn[k] > c->lastCameraCount) ? c->lastCameraCount : boundIn[k]);
            const uint32_t late = ((volatile uint32_t *)G.hCounts)[(size_t)kTableRing * kMaxSegs]; // (k_shade_sort's hint: camera rays per pass behind the root cull)
            if (late) c->lastCameraCount = late;
            const double cam = (double)nInjectedSegs * (double)(c->lastCameraCount ? c->lastCameraCount : P / 2u);
            int blocks = cam > 0.2 * others ? 3 : 4;
            if (c->tuneCorunBlocks > 0) blocks = c->tuneCorunBlocks;
            if (blocks < ct.traceBlocksPerCU) ct.traceBlocksPerCU = blocks;
        }
        launchTrace(ct, c->dScene, c->tree.leafKeys, c->tree.nodes32, c->tris, dTbl, c->dStats);
    }
    if (forked) { // (the bucket HR_KERNEL_TRACE stays k_trace's own launch; what the packet kernel beside it runs longer is booked as ray generation)
        c->timeNext(HR_KERNEL_RAYGEN, G.stream);
        HIP_TRY(c, hipStreamWaitEvent(G.stream, G.evJoin, 0));
    }
    c->timeNext(HR_KERNEL_SHADE, G.stream);
    launchShade(cfg, c->dScene, dTbl, c->dStats);
    c->timeEnd(G.stream);
    hr_ctx::PassSlot *ended[kMaxSegs];
    int nEnded = 0;
    for (int k = 0; k < n; ++k) {
        hr_ctx::PassSlot &ps = c->slots[order[k]];
        if (!c->hasPassthrough && ps.step >= ps.nIter) {
            ps.active = false, ps.finished = true;
            ended[nEnded++] = &ps;
        } else {
            ps.step++;
        }
    }
    if (nEnded > 0) HIP_TRY(c, hipEventRecord(ended[nEnded - 1]->evFinal, G.stream)); // one event for the passes whose last stage this step was
    for (int k = 0; k < nEnded; ++k) ended[k]->finalEv = ended[nEnded - 1]->evFinal;
    if (c->hasPassthrough) { // snapshot of the queue lengths after this step, read two steps from now
        uint32_t *dst = G.hQCount + (size_t)ring * kMaxSlots * kMaxBounceSlots;
        HIP_TRY(c, hipMemcpy2DAsync(dst, sizeof(uint32_t) * kMaxBounceSlots, &c->dCounters[0].qCount[0], sizeof(Counters),
                                    sizeof(uint32_t) * kMaxBounceSlots, kMaxSlots, hipMemcpyDeviceToHost, G.stream));
        HIP_TRY(c, hipEventRecord(G.statusEv[ring], G.stream));
        G.statusUsed[ring] = true;
        for (int i = 0; i < kMaxSlots; ++i)
            G.statusOrder[ring][i] = (c->slots[i].active && c->slots[i].group == g) ? c->slots[i].order + 1ull : 0ull;
    }
    HIP_TRY(c, hipGetLastError());
    if (dbgT) fprintf(stderr, "step %llu (inject %d): begin %.1f us | injected +%.1f | counts known +%.1f | table built +%.1f | copy enqueued +%.1f | launched +%.1f\n", stepIdx, nInject, tA, tB - tA, tC - tB, tD - tC, tE - tD, nowUs() - tE);
    return resolveReady(c);
}

// Stages a pass occupies in the pipeline (depth+1 shaded stages + the last occlusion stage; in pass-through scenes two more
// until the host has seen that its queue ran empty — longer only for rays that really pass through surfaces).
static int stagesOf(const hr_ctx *c, const hr_pass_params &pp) { return pp.max_ray_depth + 2 + (c->hasPassthrough ? 2 : 0); }

// Advance the group that holds the oldest in-flight pass by one macro step (keeps passes finishing in order).
static int stepOldest(hr_ctx *c)
{
    const hr_ctx::PassSlot *oldest = nullptr;
    for (const hr_ctx::PassSlot &ps : c->slots)
        if (ps.active && (!oldest || ps.order < oldest->order)) oldest = &ps;
    if (!oldest) return resolveReady(c);
    return macroStep(c, oldest->group, 0);
}

static int slotLimit(const hr_ctx *c)
{
    int limit = c->maxSlots < c->tuneDepth ? c->maxSlots : c->tuneDepth; // passes in flight, all groups
    return limit < 1 ? 1 : (limit > kMaxSlots ? kMaxSlots : limit);
}

// Inject n pending passes into the next group (round robin), first making room for them.
static int injectBatch(hr_ctx *c, int n, int perGroupLimit)
{
    const int g = c->nextGroup;
    c->nextGroup = (g + 1) % c->nGroups;
    int guard = 0;
    while ((occupiedSlots(c, g) + n > perGroupLimit || occupiedSlots(c) + n > slotLimit(c)) && occupiedSlots(c) > 0) {
        int rc = stepOldest(c);
        if (rc) return rc;
        if (++guard > 64 * kMaxBounceSlots) FAIL(c, HR_ERR_DEVICE, "internal: pass pipeline did not make room");
    }
    return macroStep(c, g, n);
}

// Passes injected together when their camera rays travel as packets: a wave holds 2^k passes of 64 >> k pixels (hr_render.hip:
// k_raygen_packets), so the batch is the power of two next to the usual one (12 -> 16, 3 -> 4, 5 -> 4), sixteen per launch at most.
static int packetBatch(const hr_ctx *c)
{
    const int b = c->injectBatch < 1 ? 1 : c->injectBatch;
    int up = 1;
    while (up < b) up <<= 1;
    return (4 * b >= 3 * up) ? up : up / 2;
}
static int packetLog2(const hr_ctx *c)
{
    int k = 0;
    while ((2 << k) <= packetBatch(c) && k < 4) ++k;
    return k;
}
static bool packetsInUse(const hr_ctx *c) { return c->tunePackets == 1 || (c->tunePackets == 2 && c->packetsOn); }

// hr_ctx_desc::memory_budget: how many passes per step fit.  A pass of the batch holds, over the `stages` steps of its life, a pass buffer
// (S + 2 of them per batch pass are kept: the pipeline's depth and the resolve lag), its camera rays and hit records (scratch), and
// what each of its closest-hit stages emits (arena: two halves, each with a third of headroom).  A stage that has not been seen yet
// counts as long as it can possibly get (one ray per owned pixel: the guarantee); a stage that has, by the largest per-pass average a
// step carved for it, plus a tenth.  All stages of a batch are in flight at once (one generation per stage), so the sum over the
// stages is what one more pass per step costs.
static double budgetBytesPerPass(const hr_ctx *c, int stages)
{
    const double P = (double)(c->queueCapacity ? c->queueCapacity : 1u), kS = c->allLightsUsed ? 4.0 : 1.0;
    const double fb = (double)c->W * c->H * 16.0 * (c->allLightsUsed ? 4.0 : 1.0);
    double arena = 0.0, scratch = 0.0;
    for (int st = 0; st + 1 < stages && st < kMaxBounceSlots; ++st) { // (the last stage traces occlusion rays only)
        arena += c->stageSeen[st] ? 1.1 * c->stageArenaSeen[st] : P * (64.0 + 48.0 * kS);
        scratch += c->stageSeen[st] ? 1.1 * c->stageScratchSeen[st] : P * 20.0 + (st == 0 ? P * 64.0 : 0.0);
    }
    return (double)c->nGroups * ((double)(stages + 2) * fb + (4.0 / 3.0) * (2.0 * arena + scratch));
}
static int budgetBatch(const hr_ctx *c, int stages)
{
    if (!c->memBudget) return 1 << 20;
    const double fit = (double)c->memBudget / budgetBytesPerPass(c, stages);
    return fit < 1.0 ? 1 : (fit > 1e6 ? 1 << 20 : (int)fit);
}

static int batchFor(const hr_ctx *c, int stages)
{
    int batch = packetsInUse(c) ? packetBatch(c) : c->injectBatch;
    const int fit = budgetBatch(c, stages);
    if (batch > fit) {
        batch = fit;
        if (packetsInUse(c)) // (a packet holds a power of two of passes)
            while (batch & (batch - 1)) batch &= batch - 1;
    }
    int perGroup = slotLimit(c) / c->nGroups;
    if (perGroup > kMaxSegs) perGroup = kMaxSegs;
    if (batch * stages > perGroup) batch = perGroup / stages;
    return batch < 1 ? 1 : batch;
}

static int drainPipeline(hr_ctx *c)
{
    while (!c->pendingInject.empty()) {
        const int stages = stagesOf(c, c->pendingInject.front());
        const int batch = batchFor(c, stages);
        int n = (int)c->pendingInject.size() < batch ? (int)c->pendingInject.size() : batch;
        // the last, partly filled batch of a run on several pipeline groups is dealt out over the groups (each group's dependent
        // chain of stages then carries a share of it, and the chains overlap on the device) instead of going to one of them whole
        if (c->nGroups > 1 && (int)c->pendingInject.size() <= batch) {
            const int idleGroups = c->nGroups - (c->nextGroup % c->nGroups);
            const int share = ((int)c->pendingInject.size() + idleGroups - 1) / (idleGroups > 0 ? idleGroups : 1);
            n = share < 1 ? 1 : share;
        }
        // With the camera rays as packets a step injects WHOLE launches of kMaxBatch passes where it can: a remainder goes in one step later,
        // where its (smaller, less coherent) packets run beside the k_trace that carries the first launches' first bounce instead of
        // lengthening the step that has nothing beside it.  A 1/8 shard's 20 passes as 16, then 4: 0.304 -> 0.294 ms/step; as 12 + 8,
        // 10 + 10, 8 + 8 + 4 (each of them smaller packets all round): 0.307 - 0.323 (profiles/r5g_burst_pmin.txt).
        if (packetsInUse(c) && n > kMaxBatch && n % kMaxBatch) n -= n % kMaxBatch;
        int perGroup = batch * stages;
        int rc = injectBatch(c, n, perGroup);
        if (rc) return rc;
    }
    int guard = 0;
    while (activePasses(c) > 0) {
        int rc = stepOldest(c);
        if (rc) return rc;
        if (++guard > 64 * kMaxBounceSlots) FAIL(c, HR_ERR_DEVICE, "internal: pass pipeline did not drain");
    }
    int rc = resolveReady(c);
    if (rc) return rc;
    if (occupiedSlots(c) > 0) FAIL(c, HR_ERR_DEVICE, "internal: finished passes left unresolved");
    c->oldestWaitingNs = 0;
    if (c->probeGuard) { // (a probe nobody has waited for: whatever follows on the caller's stream — frees after a synchronise included — comes after it)
        HIP_TRY(c, hipStreamWaitEvent(c->stream, c->evProbeB, 0));
        c->probeGuard = false;
    }
    // whatever the caller does next on its stream (clear, scene edits, new tables) has to be seen by the groups
    for (int g = 0; g < kMaxGroups; ++g) c->groups[g].needUserSync = true;
    return HR_OK;
}

int hr_render_pass(hr_ctx *c, const hr_pass_params *pp)
{
    ENTER(c);
    if (!pp) FAIL(c, HR_ERR_INVALID, "null params");
    if (c->W <= 0) FAIL(c, HR_ERR_INVALID, "no frame");
    if (!c->committed) FAIL(c, HR_ERR_INVALID, "scene not committed");
    if (c->nSeq <= 0 || c->nSeqOffsets <= 0) FAIL(c, HR_ERR_INVALID, "sample tables not set");
    if (pp->max_ray_depth < 0 || pp->max_ray_depth + 2 >= kMaxBounceSlots - 8) FAIL(c, HR_ERR_INVALID, "max_ray_depth out of range");
    if (pp->interactive_mode && (pp->block_size[0] <= 0 || pp->block_size[1] <= 0)) FAIL(c, HR_ERR_INVALID, "bad block size");
    int rc = HR_OK;
    if (pp->estimator == HR_ESTIMATOR_ENV_MIS || pp->estimator == HR_ESTIMATOR_ALL_LIGHTS) {
        rc = ensureEnvTable(c);
        if (rc) return rc;
        if (pp->estimator == HR_ESTIMATOR_ALL_LIGHTS && !c->allLightsUsed) {
            // pass slots grow (a second occlusion ray per path, a second partial sum per pass): the existing ones are released and
            // re-allocated below with the new sizes
            rc = drainPipeline(c);
            if (rc) return rc;
            QUIESCE(c);
            const uint32_t keepCap = c->queueCapacity;
            freeQueues(c);
            c->queueCapacity = keepCap;
            c->allLightsUsed = true;
            slotBudget(c);
        }
    } else if (pp->estimator != HR_ESTIMATOR_REFERENCE) {
        FAIL(c, HR_ERR_INVALID, "unknown estimator");
    }
    if (pp->texture_lod == HR_TEXTURE_LOD_CONE) {
        rc = ensureTextureLod(c);
        if (rc) return rc;
        c->textureLodUsed = true;
    } else if (pp->texture_lod != HR_TEXTURE_LOD_BASE) {
        FAIL(c, HR_ERR_INVALID, "unknown texture_lod mode");
    }
    rc = uploadScene(c); // drains the pipeline first when the scene constants changed
    if (rc) return rc;
    if (c->frame.nOwnedTiles == 0) return HR_OK;
    // only passes of equal depth overlap (keeps the groups in lockstep; order is enforced by resolveReady regardless)
    if (pp->max_ray_depth != c->lastDepth && (occupiedSlots(c) > 0 || !c->pendingInject.empty())) {
        rc = drainPipeline(c);
        if (rc) return rc;
    }
    c->lastDepth = pp->max_ray_depth;
    if (c->memBudget && (double)c->memBudget < budgetBytesPerPass(c, stagesOf(c, *pp))) {
        c->err = "hr_ctx_desc.memory_budget (" + std::to_string(c->memBudget >> 20) + " MiB) is less than one pass per pipeline step needs at " + std::to_string(c->W) + "x" +
                 std::to_string(c->H) + ", depth " + std::to_string(pp->max_ray_depth) + ": " + std::to_string((unsigned long long)budgetBytesPerPass(c, stagesOf(c, *pp)) >> 20) + " MiB";
        return HR_ERR_INVALID;
    }
    {
        // All pass slots this depth needs are allocated up front, on the first pass (hipMalloc synchronises the device and
        // takes ~0.1 ms per buffer: allocating slot by slot as the pipeline filled stalled the first 20-odd passes of a render)
        const int stagesNow = stagesOf(c, *pp);
        const int batchNow = batchFor(c, stagesNow);
        int want = c->nGroups * batchNow * stagesNow + 2 * c->nGroups * batchNow;
        if (want > slotLimit(c)) want = slotLimit(c);
        for (int i = 0; i < kMaxSlots && c->nSlotsAllocated < want; ++i)
            if (!c->slots[i].allocated) {
                rc = allocSlot(c, c->slots[i]);
                if (rc) return rc;
            }
    }
    c->pendingInject.push_back(*pp);
    if (c->oldestWaitingNs == 0)
        c->oldestWaitingNs = (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
    // a macro step is launched once enough passes are waiting to fill it; each group holds batch x stages passes
    const int stages = stagesOf(c, *pp);
    const int batch = batchFor(c, stages);
    if ((int)c->pendingInject.size() < batch) return HR_OK;
    return injectBatch(c, batch, batch * stages);
}

int hr_frame_pass_batch(hr_ctx *c, int32_t max_ray_depth, int32_t *batch)
{
    ENTER(c);
    if (!batch || max_ray_depth < 0) FAIL(c, HR_ERR_INVALID, "bad arguments");
    if (c->W <= 0) FAIL(c, HR_ERR_INVALID, "no frame");
    hr_pass_params pp{};
    pp.max_ray_depth = max_ray_depth;
    *batch = batchFor(c, stagesOf(c, pp));
    return HR_OK;
}

int hr_flush(hr_ctx *c)
{
    ENTER(c);
    const int rc = drainPipeline(c);
    return rc ? rc : overflowCheck(c); // (no wait here: what the kernels have reported so far)
}

int hr_get_stats(hr_ctx *c, hr_pass_stats *out)
{
    ENTER(c);
    if (!out) FAIL(c, HR_ERR_INVALID, "null output");
    {
        int rc = drainPipeline(c);
        if (rc) return rc;
    }
    std::vector<Stats> parts(kStatSlots);
    HIP_TRY(c, hipMemcpyAsync(parts.data(), c->dStats, sizeof(Stats) * kStatSlots, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    {
        const int rc = overflowCheck(c);
        if (rc) return rc;
    }
    Stats s{};
    for (const Stats &p : parts) {
        s.paths += p.paths, s.raysClosest += p.raysClosest, s.raysAny += p.raysAny, s.shadedHits += p.shadedHits;
        s.accumulates += p.accumulates, s.nodeVisits += p.nodeVisits, s.triTests += p.triTests;
        s.nodeVisitsAny += p.nodeVisitsAny, s.triTestsAny += p.triTestsAny;
    }
    std::memset(out, 0, sizeof(*out));
    out->paths = s.paths, out->rays_closest = s.raysClosest, out->rays_any = s.raysAny, out->shaded_hits = s.shadedHits;
    out->accumulates = s.accumulates, out->node_visits = s.nodeVisits, out->tri_tests = s.triTests;
    out->node_visits_any = s.nodeVisitsAny, out->tri_tests_any = s.triTestsAny;
    return HR_OK;
}

int hr_get_kernel_times(hr_ctx *c, hr_kernel_times *out)
{
    ENTER(c);
    if (!out) FAIL(c, HR_ERR_INVALID, "null output");
    {
        int rc = drainPipeline(c);
        if (rc) return rc;
    }
    c->drainTimes();
    for (int k = 0; k < HR_KERNEL_COUNT; ++k) out->ms[k] = c->kernelMs[k], out->launches[k] = c->kernelLaunches[k];
    std::vector<Stats> parts(kStatSlots);
    HIP_TRY(c, hipMemcpyAsync(parts.data(), c->dStats, sizeof(Stats) * kStatSlots, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    unsigned long long ticks = 0, launches = 0;
    for (const Stats &p : parts) ticks += p.traceTicks, launches += p.traceLaunches;
    out->trace_clock_ms = (float)((double)ticks * 1e-5); // 100 MHz: 10 ns per tick
    out->trace_clock_launches = (uint32_t)launches;
    out->camera_packets = packetsInUse(c) ? (uint32_t)packetBatch(c) : 0u;
    out->packet_union = (float)c->lastUnion;
    return HR_OK;
}

int hr_get_step_log(hr_ctx *c, hr_step_record *out, int32_t capacity, int32_t *n_records)
{
    ENTER(c);
    if (!out || !n_records || capacity <= 0) FAIL(c, HR_ERR_INVALID, "bad arguments");
    {
        int rc = drainPipeline(c);
        if (rc) return rc;
    }
    std::vector<Stats> parts(1);
    std::vector<unsigned long long> log(3 * (size_t)kStepLogCap);
    HIP_TRY(c, hipMemcpyAsync(parts.data(), c->dStats, sizeof(Stats), hipMemcpyDeviceToHost, c->stream)); // (k_shade_sort's first thread counts in the first copy)
    HIP_TRY(c, hipMemcpyAsync(log.data(), c->dStepLog, log.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const unsigned long long total = parts[0].traceLaunches;
    const unsigned long long first = total > (unsigned long long)kStepLogCap ? total - (unsigned long long)kStepLogCap : 0ull;
    // Records are appended when a step's k_trace has ENDED (k_shade_sort writes them), so with several pipeline groups they arrive out of
    // start order: they are handed out sorted by start, each with its group.
    std::vector<const unsigned long long *> recs;
    for (unsigned long long i = first; i < total; ++i) recs.push_back(&log[3 * (size_t)(i % (unsigned long long)kStepLogCap)]);
    std::stable_sort(recs.begin(), recs.end(), [](const unsigned long long *a, const unsigned long long *b) { return a[0] < b[0]; });
    int32_t n = 0;
    for (const unsigned long long *rec : recs) {
        if (n >= capacity) break;
        out[n].start_ms = (double)(rec[0] - recs[0][0]) * 1e-5; // 100 MHz device clock
        out[n].trace_ms = (float)((double)(rec[1] - rec[0]) * 1e-5);
        out[n].passes_in_flight = (int32_t)(rec[2] & 0xFFFFull), out[n].group = (int32_t)((rec[2] >> 16) & 0xFFull), out[n].passes_injected = (int32_t)(rec[2] >> 32);
        ++n;
    }
    *n_records = n;
    return HR_OK;
}

