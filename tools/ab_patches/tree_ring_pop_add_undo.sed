# round 3's ring_pop (add to head, undo on failure) inside this round's tree_kernels.hip: the A/B partner of the credit protocol for
# c6_pool_ring_selftest (ADVICE r3: the undo could hand one ring slot to two poppers).  tools/ab_patch.sh old_ring tree_kernels.hip tools/ab_patches/tree_ring_pop_add_undo.sed
/^            h = atomicAdd(&ctr\[head_i\], (unsigned long long)n);$/d
s|^        const long long a = (long long)atomicAdd(&ctr\[avail_i\], 0ull - (unsigned long long)n);$|        h = atomicAdd(\&ctr[head_i], (unsigned long long)n); const long long a = (long long)(ctr[head_i == C6_POOL_PAGE_HEAD ? C6_POOL_PAGE_LIMIT : C6_POOL_CHUNK_LIMIT] - h);|
s|^        if (!ok) atomicAdd(&ctr\[avail_i\], (unsigned long long)n);$|        if (!ok) atomicAdd(\&ctr[head_i], 0ull - (unsigned long long)n);|
