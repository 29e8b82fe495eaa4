s|const float puct = prob \* sqrtf(visit_sum - visit) / (1.f + visit);|const float puct = prob * __builtin_amdgcn_sqrtf(visit_sum - visit) * __builtin_amdgcn_rcpf(1.f + visit);|
s|const float qv = qs / (float)v;   // recalc_q :77-79|const float qv = qs * __builtin_amdgcn_rcpf((float)v);|
