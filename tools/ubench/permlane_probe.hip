// permlane_probe.hip -- what does v_permlane32_swap_b32 do, lane by lane?  (gfx950; used by k_synthesize_eo_team's hop-128 accumulator shift)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/permlane_probe.hip -o tools/ubench/permlane_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k( unsigned * out )
	{
	const unsigned lane = threadIdx.x;
	const unsigned a = lane, b = 100 + lane;
	auto r = __builtin_amdgcn_permlane32_swap( a, b, false, false );
	out[lane] = r[0];
	out[64 + lane] = r[1];
	}
int main()
	{
	unsigned * d = nullptr, h[128];
	if( hipMalloc( &d, sizeof( h ) ) != hipSuccess ) return 1;
	hipLaunchKernelGGL( k, dim3( 1 ), dim3( 64 ), 0, 0, d );
	if( hipMemcpy( h, d, sizeof( h ), hipMemcpyDeviceToHost ) != hipSuccess ) return 1;
	printf( "permlane32_swap( a = lane, b = 100 + lane ):\n r[0]: lane0 %u lane31 %u lane32 %u lane63 %u\n r[1]: lane0 %u lane31 %u lane32 %u lane63 %u\n",
		h[0], h[31], h[32], h[63], h[64], h[95], h[96], h[127] );
	return 0;
	}
