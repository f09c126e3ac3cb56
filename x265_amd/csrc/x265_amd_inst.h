// The heavy kernel instantiations (general search, predictor-chained field, lookahead), listed once:
// x265_amd.hip declares them `extern template` and x265_amd_part.hip defines one group of them per
// translation unit (-DXA_PART=1..7), so the groups compile side by side (x265_amd/build.py).
// XA_INST is `extern template` or `template`; XA_PART (optional) selects a group.
#define XA_K_SEARCH(PIX, NW, MAXG, LR, INL)                                                                              \
    XA_INST __global__ void xa::k_search<xa::JobLds<PIX, NW, MAXG, LR, INL>>(xa::BatchParams, const x265amd_me_job*, const int*, \
                                                                           int, int, const xa::FullBest*, int,           \
                                                                           x265amd_me_result*, unsigned long long*);
#define XA_K_SEARCH_BIG(PIX)                                                                                             \
    XA_K_SEARCH(PIX, 8, 1024, false, true) XA_K_SEARCH(PIX, 8, 1024, true, false)                                        \
    XA_K_SEARCH(PIX, 4, 1024, false, true) XA_K_SEARCH(PIX, 4, 1024, true, false)
#define XA_K_SEARCH_SMALL(PIX)                                                                                           \
    XA_K_SEARCH(PIX, 1, 256, false, true) XA_K_SEARCH(PIX, 1, 64, false, true)                                           \
    XA_K_SEARCH(PIX, 1, 256, true, false) XA_K_SEARCH(PIX, 1, 64, true, false)
#define XA_K_FIELD(PIX, NW, MAXG)                                                                                        \
    XA_INST __global__ void xa::k_field<xa::JobLds<PIX, NW, MAXG, false, NW == 1, NW == 1>>(const xa::PlaneDesc*, const uint16_t*, \
                                                                                 const xa::FieldDesc*, int, int, int, int, int, int, \
                                                                                 int, int, int, const x265amd_me_result*, x265amd_me_result*, \
                                                                                 int*, int*, unsigned long long*);
#define XA_K_LOOK(PIX)                                                                                                   \
    XA_INST __global__ void xa::k_lookahead<xa::JobLds<PIX, 1, 64, true, true>>(const xa::PlaneDesc*, const uint16_t*,    \
                                                                              const xa::LookPair*, int, int, int, int, int, int, int, \
                                                                              x265amd_me_result*, int*, unsigned long long*);
#define XA_K_LANE(PIX, LPJ)                                                                                              \
    XA_INST __global__ void xa::k_lane<PIX, LPJ>(xa::BatchParams, const x265amd_me_job*, const int*, int, const xa::FullBest*, int, \
                                                 x265amd_me_result*, unsigned long long*);
#if !defined(XA_PART) || XA_PART == 6
XA_K_LANE(uint8_t, 4) XA_K_LANE(uint16_t, 4)
#endif
#if !defined(XA_PART) || XA_PART == 7
XA_K_LANE(uint8_t, 16) XA_K_LANE(uint16_t, 16)
#endif
#if !defined(XA_PART) || XA_PART == 1
XA_K_SEARCH_BIG(uint8_t)
#endif
#if !defined(XA_PART) || XA_PART == 2
XA_K_SEARCH_SMALL(uint8_t)
#endif
#if !defined(XA_PART) || XA_PART == 3
XA_K_SEARCH_BIG(uint16_t)
#endif
#if !defined(XA_PART) || XA_PART == 4
XA_K_SEARCH_SMALL(uint16_t)
#endif
#if !defined(XA_PART) || XA_PART == 5
XA_K_FIELD(uint8_t, 1, 64) XA_K_FIELD(uint8_t, 1, 256) XA_K_FIELD(uint8_t, 8, 1024)
XA_K_FIELD(uint16_t, 1, 64) XA_K_FIELD(uint16_t, 1, 256) XA_K_FIELD(uint16_t, 8, 1024)
XA_K_LOOK(uint8_t) XA_K_LOOK(uint16_t)
#endif
#undef XA_K_SEARCH
#undef XA_K_SEARCH_BIG
#undef XA_K_SEARCH_SMALL
#undef XA_K_FIELD
#undef XA_K_LOOK
#undef XA_K_LANE
