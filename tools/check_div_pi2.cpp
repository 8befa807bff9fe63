// Exhaustive proof that q = fma(fma(-x*rc, c, x), rc, x*rc) equals the IEEE quotient x / c for c = float pi2 and every
// float |x| >= 1e-30 (flan_amd/csrc/pv_math.h div_pi2).  g++ -O2 -ffp-contract=off -mfma tools/check_div_pi2.cpp -lpthread && ./a.out  (~30 s on 8 cores)
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>
#include <atomic>
int main(){
  const float c = 6.2831854820251465f;
  const float rc = 1.0f / c;
  std::atomic<uint64_t> bad(0), total(0);
  std::vector<std::thread> th;
  for(int t=0;t<8;++t) th.emplace_back([&,t]{
    uint64_t lb=0, lt=0;
    for(uint64_t u = (uint64_t)t<<29; u < ((uint64_t)(t+1)<<29); ++u){
      uint32_t b=(uint32_t)u; float x; memcpy(&x,&b,4);
      if(!std::isfinite(x) || std::fabs(x) < 1e-30f) continue;
      float q0 = x*rc; float r = fmaf(-q0,c,x); float q = fmaf(r,rc,q0);
      float ref = x / c;
      uint32_t a1,a2; memcpy(&a1,&q,4); memcpy(&a2,&ref,4);
      ++lt; if(a1!=a2){ if(lb<3) printf("x=%a q=%a ref=%a\n",x,q,ref); ++lb; }
    }
    bad+=lb; total+=lt;
  });
  for(auto&x:th) x.join();
  printf("rc=%a bad=%llu of %llu\n", rc, (unsigned long long)bad.load(), (unsigned long long)total.load());
}
