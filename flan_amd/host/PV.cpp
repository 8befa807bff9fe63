// PV.cpp -- conversions and frame processors of flan::PV over the C ABI
// (reference: Conversions/AudioPV.cpp:86-145, PV/PVModify.cpp:196-385, PV/PV.cpp:421-458).
#include "flan/PV.h"

#include <algorithm>
#include <cmath>
#include <limits>
#include <iostream>

#include "device_block.h"
#include "flan/Audio.h"

namespace flan {

namespace {
using detail::DeviceBlock;

std::shared_ptr<DeviceBlock> upload( const void * host, size_t bytes )
	{
	auto b = DeviceBlock::allocate( bytes );
	if( !b ) return nullptr;
	if( !detail::report( flanhip_memcpy_h2d( b->ptr, host, bytes, nullptr ), "upload" ) ) return nullptr;
	if( !detail::report( flanhip_stream_synchronize( nullptr ), "upload" ) ) return nullptr;   // a staging block goes back to its cache after this
	return b;
	}

// Declared AFTER the staging vectors / device blocks of a function that starts asynchronous copies, so that it is destroyed FIRST: on
// every way out -- an early `return nullptr`, an exception thrown by the user's callable (pool_run rethrows it) -- the copy streams
// and the null stream are drained before page-locked and device blocks go back to their caches, where the next user could receive them
// while a DMA still reads or writes them.
struct DmaQuiesce
	{
	detail::CopyStreams streams;
	explicit DmaQuiesce( detail::CopyStreams s = detail::CopyStreams() ) : streams( s ) {}
	DmaQuiesce( const DmaQuiesce & ) = delete;
	DmaQuiesce & operator=( const DmaQuiesce & ) = delete;
	~DmaQuiesce()
		{
		if( streams.down ) (void) flanhip_stream_synchronize( streams.down );
		if( streams.up ) (void) flanhip_stream_synchronize( streams.up );
		(void) flanhip_stream_synchronize( nullptr );
		}
	};

// sample_function_over_domain (PV.h:31-35) straight onto the device: a constant is filled there; a callable is sampled on the
// host a slab of frames at a time into page-locked memory, each slab on its way over the link while the next is sampled.
// `keep`, when given, receives the host copy (modify_time needs its maximum).
template<typename T>
std::shared_ptr<DeviceBlock> function_grid_to_device( const PV & me, const Function<TF, T> & f, FunctionSample2d<T> * keep = nullptr, Frame num_frames = -1 )
	{
	static_assert( std::is_trivially_copyable_v<T>, "plain grids only" );
	const int frames = num_frames >= 0 ? num_frames : me.get_num_frames(), bins = me.get_num_bins();   // num_frames: a domain with this PV's rates but another length
	const size_t count = size_t( frames ) * bins;
	if( f.is_constant() )
		{
		auto b = DeviceBlock::allocate( sizeof( T ) * count );
		if( !b ) return nullptr;
		if constexpr( std::is_same_v<T, float> )
			{
			if( !detail::report( flanhip_fill_dev( static_cast<float*>( b->ptr ), int64_t( count ), float( f.get_constant() ), nullptr ), "fill" ) ) return nullptr;
			}
		else
			{
			detail::StagingVector<T> host( count );
			std::fill( host.begin(), host.end(), f.get_constant() );
			if( !detail::report( flanhip_memcpy_h2d( b->ptr, host.data(), sizeof( T ) * count, nullptr ), "upload" ) ) return nullptr;
			if( !detail::report( flanhip_stream_synchronize( nullptr ), "upload" ) ) return nullptr;
			}
		if( keep ) *keep = FunctionSample2d<T>{ f.get_constant(), count, size_t( bins ) };
		return b;
		}
	typename FunctionSample2d<T>::Vector host( count );
	auto b = DeviceBlock::allocate( sizeof( T ) * count );
	if( !b ) return nullptr;
	const DmaQuiesce quiesce;                                                     // the uploads below are asynchronous on the null stream: drained on every way out
	const int slabs = count * sizeof( T ) >= ( size_t( 4 ) << 20 ) ? 4 : 1;
	for( int k = 0; k < slabs; ++k )
		{
		const int x0 = int( int64_t( frames ) * k / slabs ), x1 = int( int64_t( frames ) * ( k + 1 ) / slabs );
		if( x1 <= x0 ) continue;
		f.sample_into( host.data() + size_t( x0 ) * bins, float( x0 ), float( x1 ), 1.0f / me.get_analysis_rate(), 0, float( bins ), me.bin_to_frequency( 1 ) );
		if( !detail::report( flanhip_memcpy_h2d( static_cast<char*>( b->ptr ) + sizeof( T ) * size_t( x0 ) * bins, host.data() + size_t( x0 ) * bins,
				sizeof( T ) * size_t( x1 - x0 ) * bins, nullptr ), "upload" ) ) return nullptr;
		}
	if( !detail::report( flanhip_stream_synchronize( nullptr ), "upload" ) ) return nullptr;   // before the staging block goes back to its cache
	if( keep ) *keep = FunctionSample2d<T>{ std::move( host ), count, size_t( bins ) };
	return b;
	}

// A callable that must see the PV's own data (modify_frequency's mod at every MF's frequency, PVModify.cpp:263-268; shape's
// shaper, PV.cpp:435-436) runs on the host: body( row, row's MFs, row's outputs ) for every (channel, frame) row, results on the
// device.  When the data lives on the device only it is NOT brought into the PV's host vector: it comes over in slabs of rows
// through page-locked memory, slab k+1 on its way down and slab k-1 on its way up while slab k is evaluated.
template<typename Out, typename Body>
std::shared_ptr<DeviceBlock> map_rows_to_device( const PV & me, ExecutionPolicy policy, const Body & body )
	{
	const int rows = int( size_t( me.get_num_channels() ) * me.get_num_frames() ), bins = me.get_num_bins();
	const size_t count = size_t( rows ) * bins;
	auto d_out = DeviceBlock::allocate( sizeof( Out ) * count );
	if( !d_out ) return nullptr;
	detail::StagingVector<Out> out( count );
	const bool on_host = me.host_copy_is_current();
	const MF * d_in = on_host ? nullptr : me.device_data();
	if( !on_host && !d_in ) return nullptr;
	detail::StagingVector<MF> in( on_host ? 0 : count );
	const MF * src = on_host ? me.get_buffer().data() : in.data();
	const int slabs = std::max( 1, std::min<int>( 16, int( count * sizeof( MF ) >> 22 ) ) );
	const detail::CopyStreams streams = detail::copy_streams();                  // downloads and uploads on a stream each: both directions of the link at once
	const DmaQuiesce quiesce( streams );                                         // last declared, first destroyed: no copy outlives `in`, `out`, `d_out`
	auto slab_begin = [&]( int k ){ return int( int64_t( rows ) * k / slabs ); };
	auto fetch = [&]( int k )
		{
		if( on_host || k >= slabs ) return true;
		const size_t lo = size_t( slab_begin( k ) ) * bins, hi = size_t( slab_begin( k + 1 ) ) * bins;
		return hi <= lo || detail::report( flanhip_memcpy_d2h( in.data() + lo, d_in + lo, sizeof( MF ) * ( hi - lo ), streams.down ), "download" );
		};
	if( !detail::report( flanhip_stream_synchronize( nullptr ), "synchronise" ) ) return nullptr;   // whatever produced the data has finished
	if( !fetch( 0 ) || !detail::report( flanhip_stream_synchronize( streams.down ), "download" ) ) return nullptr;
	bool ok = true;
	for( int k = 0; k < slabs && ok; ++k )
		{
		ok = fetch( k + 1 );
		const int r0 = slab_begin( k ), r1 = slab_begin( k + 1 );
		detail::for_each_index( r0, r1, policy, [&]( int row ){ body( row, src + size_t( row ) * bins, out.data() + size_t( row ) * bins ); }, 16 );
		const size_t lo = size_t( r0 ) * bins, hi = size_t( r1 ) * bins;
		if( ok && hi > lo ) ok = detail::report( flanhip_memcpy_h2d( static_cast<Out*>( d_out->ptr ) + lo, out.data() + lo, sizeof( Out ) * ( hi - lo ), streams.up ), "upload" );
		ok = detail::report( flanhip_stream_synchronize( streams.down ), "download" ) && ok;       // slab k+1 is here
		}
	ok = detail::report( flanhip_stream_synchronize( streams.up ), "upload" ) && ok;               // before the staging blocks go back to their cache
	if( !ok ) return nullptr;
	return d_out;
	}

// FunctionSample::maximum (FunctionSample.h:156-160) = std::max_element, over the rows in parallel.  Its answer with NaNs in
// the grid: the first element if that is a NaN, else the largest of the others.
float grid_maximum( const FunctionSample2d<float> & s, ExecutionPolicy policy )
	{
	if( s.is_constant() ) return s.get_constant();
	const auto & v = s.get_vector();
	if( v.empty() || v[0] != v[0] ) return v.empty() ? 0.0f : v[0];
	const size_t row = std::max<size_t>( s.small_dim_size, 1 ), rows = ( v.size() + row - 1 ) / row;
	std::vector<float> row_max( rows );
	detail::for_each_index( 0, int( rows ), policy, [&]( int r )
		{
		float mx = -INFINITY;
		const size_t hi = std::min( v.size(), ( size_t( r ) + 1 ) * row );
		for( size_t i = size_t( r ) * row; i < hi; ++i ) mx = v[i] > mx ? v[i] : mx;
		row_max[size_t( r )] = mx;
		}, 64 );
	return *std::max_element( row_max.begin(), row_max.end() );
	}

}

PV::PV() : PVBuffer( PVBuffer::Format() ) {}
PV::PV( PVBuffer && other ) : PVBuffer( std::move( other ) ) {}
PV PV::create_null() { return PVBuffer(); }
PV PV::create_from_format( const PVBuffer::Format & f ) { return PVBuffer( f ); }
PV PV::load_from_file( const std::string & filename ) { return PVBuffer( filename ); }
PV PV::copy() const { return PVBuffer::copy(); }

// PV.cpp:41-90
MF PV::getBinInterpolated( Channel channel, fFrame frame, fBin bin, const Interpolator & i ) const
	{
	const MF p0 = get_MF( channel, Frame( std::floor( frame ) ), Bin( std::floor( bin ) ) ), p1 = get_MF( channel, Frame( std::ceil( frame ) ), Bin( std::floor( bin ) ) );
	const MF p2 = get_MF( channel, Frame( std::ceil( frame ) ), Bin( std::ceil( bin ) ) ), p3 = get_MF( channel, Frame( std::floor( frame ) ), Bin( std::ceil( bin ) ) );
	const float l = i( frame - std::floor( frame ) ), m = i( bin - std::floor( bin ) );
	const float mI = 1.0f - m, lI = 1.0f - l;
	return MF{ mI * ( lI * p0.m + l * p1.m ) + m * ( lI * p3.m + l * p2.m ), mI * ( lI * p0.f + l * p1.f ) + m * ( lI * p3.f + l * p2.f ) };
	}

MF PV::getBinInterpolated( Channel channel, fFrame frame, Bin bin, const Interpolator & i ) const
	{
	const MF l = get_MF( channel, Frame( std::floor( frame ) ), bin ), h = get_MF( channel, Frame( std::ceil( frame ) ), bin );
	const float mix = i( frame - std::floor( frame ) );
	return MF{ ( 1.0f - mix ) * l.m + mix * h.m, ( 1.0f - mix ) * l.f + mix * h.f };
	}

MF PV::getBinInterpolated( Channel channel, Frame frame, fBin bin, const Interpolator & i ) const
	{
	const MF l = get_MF( channel, frame, Bin( std::floor( bin ) ) ), h = get_MF( channel, frame, Bin( std::ceil( bin ) ) );
	const float mix = i( bin - std::floor( bin ) );
	return MF{ ( 1.0f - mix ) * l.m + mix * h.m, ( 1.0f - mix ) * l.f + mix * h.f };
	}

Audio PV::convert_to_audio( flan_CANCEL_ARG_CPP ) const
	{
	if( is_null() ) return Audio::create_null();
	if( canceller ) return Audio::create_null();               // flan_CANCEL_POINT, AudioPV.cpp:115
	AudioBuffer::Format af;                                    // AudioPV.cpp:91-94
	af.num_channels = get_num_channels();
	af.num_frames = get_num_frames() * get_hop_size();
	af.sample_rate = get_sample_rate();
	if( get_hop_size() < 1 ) return Audio::create_null();

	const MF * d_pv = device_data();
	if( !d_pv ) return Audio::create_null();
	const size_t ws_bytes = flanhip_synthesize_workspace_bytes( get_num_channels(), get_num_frames(), get_num_bins(), get_sample_rate(), get_analysis_rate(), get_window_size() );
	if( ws_bytes == 0 ) { detail::report( FLANHIP_ERR_UNSUPPORTED, "convert_to_audio (unsupported dft / window)" ); return Audio::create_null(); }
	auto out = DeviceBlock::allocate( sizeof( float ) * size_t( af.num_channels ) * af.num_frames );
	const bool conditional = synthesis_workspace_is_conditional();
	auto fused_ws = take_synthesis_workspace();                 // left by convert_to_PV / modify_time when this PV came straight from them
	const bool fused = fused_ws && fused_ws->bytes >= ws_bytes;
	auto ws = fused ? fused_ws : DeviceBlock::allocate( ws_bytes );
	auto flag = DeviceBlock::allocate( sizeof( int ) );
	if( !out || !ws || !flag ) return Audio::create_null();
	flanhip_memset( flag->ptr, 0, sizeof( int ), nullptr );
	if( canceller ) return Audio::create_null();
	const auto fused_entry = conditional ? flanhip_synthesize_dev_fused_checked : flanhip_synthesize_dev_fused;
	const int rc = fused
		? fused_entry( reinterpret_cast<const flanhip_MF*>( d_pv ), get_num_channels(), get_num_frames(), get_num_bins(),
			get_sample_rate(), get_analysis_rate(), get_window_size(), static_cast<float*>( out->ptr ), ws->ptr, static_cast<int*>( flag->ptr ), nullptr )
		: flanhip_synthesize_dev( reinterpret_cast<const flanhip_MF*>( d_pv ), get_num_channels(), get_num_frames(), get_num_bins(),
			get_sample_rate(), get_analysis_rate(), get_window_size(), static_cast<float*>( out->ptr ), ws->ptr, static_cast<int*>( flag->ptr ), nullptr );
	if( !detail::report( rc, "convert_to_audio" ) ) return Audio::create_null();
	int nan_flag = 0;
	flanhip_memcpy_d2h( &nan_flag, flag->ptr, sizeof( int ), nullptr );
	const int waited = flanhip_wait_cancellable_fn( nullptr, detail::poll_canceller, &canceller );   // flan_CANCEL_POINT while the kernels run
	if( waited == FLANHIP_ERR_CANCELLED ) return Audio::create_null();
	if( !detail::report( waited, "convert_to_audio" ) ) return Audio::create_null();
	if( nan_flag & 2 ) { detail::report( FLANHIP_ERR_INVALID_ARG, "convert_to_audio (the synthesis workspace was overwritten by another producer)" ); return Audio::create_null(); }
	if( nan_flag & 1 )                                         // AudioPV.cpp:88-89
		std::cout << "flan::convert_to_audio recieved a nan or infinite value. This often happens when dividing by zero in an earlier algorithm.";
	if( canceller ) return Audio::create_null();
	return AudioBuffer::adopt_device( af, std::move( out ) );
	}

Audio PV::convertToAudio( flan_CANCEL_ARG_CPP ) const { return convert_to_audio( canceller ); }

Audio PV::convert_to_lr_audio( flan_CANCEL_ARG_CPP ) const
	{
	if( get_num_channels() != 2 ) return Audio::create_null(); // AudioPV.cpp:143
	return convert_to_audio( canceller ).convert_to_left_right();
	}

// What the kernels need of an Interpolator: the named ones (Utility/Interpolator.cpp:14-101) are evaluated on the device by kind; one built
// from a user's callable is sampled here at i / 65536 and registered as a table the kernels read with linear interpolation between the
// samples (include/flanhip.h: flanhip_interp_table_create) -- the reference calls the callable from its parallel loops at the same kind of
// argument, a position in [0, 1] inside a frame or bin pair (PVModify.cpp:232, :344, :491).  Sampled on the calling thread.
namespace {
class DeviceInterp
	{
public:
	explicit DeviceInterp( const Interpolator & interp )
		{
		if( interp.kind() >= 0 ) { kind_ = interp.kind(); return; }
		std::vector<float> table( size_t( FLANHIP_INTERP_TABLE_INTERVALS ) + 2 );
		for( int i = 0; i <= FLANHIP_INTERP_TABLE_INTERVALS; ++i ) table[size_t( i )] = interp( float( i ) * ( 1.0f / float( FLANHIP_INTERP_TABLE_INTERVALS ) ) );
		table.back() = interp( std::numeric_limits<float>::quiet_NaN() );
		int kind = -1;
		if( detail::report( flanhip_interp_table_create( table.data(), &kind ), "interpolator table" ) ) { kind_ = kind; owned_ = true; }
		}
	~DeviceInterp() { if( owned_ ) (void) flanhip_interp_table_destroy( kind_ ); }
	DeviceInterp( const DeviceInterp & ) = delete;
	DeviceInterp & operator=( const DeviceInterp & ) = delete;
	bool ok() const { return kind_ >= 0; }
	int kind() const { return kind_; }
private:
	int kind_ = -1;
	bool owned_ = false;
	};
} // namespace

// modify_time_base, PVModify.cpp:307-362
static PV modify_time_device( const PV & me, std::shared_ptr<DeviceBlock> d_mod, float max_seconds, int interp )
	{
	const float last_output_frame = std::ceil( me.time_to_frame( max_seconds ) );    // :312
	PVBuffer::Format f = me.get_format();
	f.num_frames = Frame( last_output_frame );                                     // :315
	if( f.num_frames <= 0 ) return PV();
	const MF * d_pv = me.device_data();
	auto out = DeviceBlock::allocate( sizeof( MF ) * size_t( f.num_channels ) * f.num_frames * f.num_bins );
	if( !d_pv || !out ) return PV();
	// the kernel that writes the output can also leave convert_to_audio's pre-pass in a workspace for the new PV (it can when the
	// time map never runs backwards -- decided on the device): hand that workspace to the result
	const size_t ws_bytes = flanhip_synthesize_workspace_bytes( f.num_channels, f.num_frames, f.num_bins, f.sample_rate, f.analysis_rate, f.window_size );
	auto ws = ws_bytes ? DeviceBlock::allocate( ws_bytes ) : nullptr;
	const int rc = ws
		? flanhip_modify_time_interp_dev_fused( reinterpret_cast<const flanhip_MF*>( d_pv ), me.get_num_channels(), me.get_num_frames(), me.get_num_bins(),
			me.get_sample_rate(), me.get_analysis_rate(), static_cast<const float*>( d_mod->ptr ), f.num_frames, interp, static_cast<flanhip_MF*>( out->ptr ),
			me.get_window_size(), ws->ptr, nullptr )
		: flanhip_modify_time_interp_dev( reinterpret_cast<const flanhip_MF*>( d_pv ), me.get_num_channels(), me.get_num_frames(), me.get_num_bins(),
			me.get_sample_rate(), me.get_hop_size(), static_cast<const float*>( d_mod->ptr ), f.num_frames, interp, static_cast<flanhip_MF*>( out->ptr ), nullptr );
	if( !detail::report( rc, "modify_time" ) ) return PV();
	if( !detail::report( flanhip_stream_synchronize( nullptr ), "modify_time" ) ) return PV();
	PV result = PVBuffer::adopt_device( f, std::move( out ) );
	if( ws ) result.attach_synthesis_workspace( std::move( ws ), true );
	return result;
	}

PV PV::modify_time( const Function<TF, Second> & mod, const Interpolator & interp ) const
	{
	if( is_null() ) return PV();
	const DeviceInterp device_interp( interp );                                      // named kind, or the callable sampled into a table
	if( !device_interp.ok() ) return PV();
	FunctionSample2d<Second> sampled{ 0.0f, 0, 0 };
	auto d_mod = function_grid_to_device( *this, mod, &sampled );                  // PVModify.cpp:367
	const float mx = grid_maximum( sampled, mod.get_execution_policy() );          // FunctionSample::maximum
	if( !d_mod ) return PV();
	return modify_time_device( *this, std::move( d_mod ), mx, device_interp.kind() );
	}

PV PV::stretch( const Function<TF, float> & factor, const Interpolator & interp ) const
	{
	if( is_null() ) return PV();
	const DeviceInterp device_interp( interp );                                      // named kind, or the callable sampled into a table
	if( !device_interp.ok() ) return PV();
	auto d_max = DeviceBlock::allocate( sizeof( float ) );
	std::shared_ptr<DeviceBlock> d_grid;
	if( factor.is_constant() )
		{
		// a constant factor: the map from the number alone (the running sum of a constant in closed form, csrc/const_sum.h) -- no grid, no scan
		d_grid = DeviceBlock::allocate( sizeof( float ) * size_t( get_num_frames() ) * get_num_bins() );
		if( !d_grid || !d_max ) return PV();
		if( !detail::report( flanhip_stretch_map_const_dev( factor.get_constant(), static_cast<float*>( d_grid->ptr ), get_num_frames(), get_num_bins(), get_sample_rate(),
				get_hop_size(), static_cast<float*>( d_max->ptr ), nullptr ), "stretch" ) ) return PV();
		}
	else
		{
		d_grid = function_grid_to_device( *this, factor );                           // PVModify.cpp:373
		if( !d_grid || !d_max ) return PV();
		// :376-382 running sum over frames per bin, frame_to_time -- on the device, plus the maximum modify_time_base needs
		if( !detail::report( flanhip_stretch_map_dev( static_cast<float*>( d_grid->ptr ), get_num_frames(), get_num_bins(), get_sample_rate(), get_hop_size(),
				static_cast<float*>( d_max->ptr ), nullptr ), "stretch" ) ) return PV();
		}
	float mx = 0.0f;
	flanhip_memcpy_d2h( &mx, d_max->ptr, sizeof( float ), nullptr );
	if( !detail::report( flanhip_stream_synchronize( nullptr ), "stretch" ) ) return PV();
	return modify_time_device( *this, std::move( d_grid ), mx, device_interp.kind() );
	}

// modify_frequency_base, PVModify.cpp:196-257
static PV modify_frequency_device( const PV & me, const DeviceBlock & d_mod, const DeviceBlock & d_in_modified, int interp )
	{
	const MF * d_pv = me.device_data();
	const size_t n = size_t( me.get_num_channels() ) * me.get_num_frames() * me.get_num_bins();
	auto out = DeviceBlock::allocate( sizeof( MF ) * n );
	if( !d_pv || !out ) return PV();
	if( !detail::report( flanhip_modify_frequency_interp_dev( reinterpret_cast<const flanhip_MF*>( d_pv ), me.get_num_channels(), me.get_num_frames(), me.get_num_bins(),
			me.get_sample_rate(), static_cast<const float*>( d_mod.ptr ), static_cast<const float*>( d_in_modified.ptr ), interp, static_cast<flanhip_MF*>( out->ptr ), nullptr ),
			"modify_frequency" ) ) return PV();
	if( !detail::report( flanhip_stream_synchronize( nullptr ), "modify_frequency" ) ) return PV();
	return PVBuffer::adopt_device( me.get_format(), std::move( out ) );
	}

PV PV::modify_frequency( const Function<TF, Frequency> & mod, const Interpolator & interp ) const
	{
	if( is_null() ) return PV();
	const DeviceInterp device_interp( interp );                                      // named kind, or the callable sampled into a table
	if( !device_interp.ok() ) return PV();
	auto d_mod = function_grid_to_device( *this, mod );                            // PVModify.cpp:261
	if( !d_mod ) return PV();
	// :263-268: the callable is evaluated at every MF's own (time, frequency): data dependent, so on the host
	auto d_in = map_rows_to_device<float>( *this, mod.get_execution_policy(), [&]( int row, const MF * mfs, float * out )
		{
		const Second t = frame_to_time( fFrame( Frame( row % get_num_frames() ) ) );
		for( Bin bin = 0; bin < get_num_bins(); ++bin ) out[bin] = mod( TF{ t, mfs[bin].f } );
		} );
	if( !d_in ) return PV();
	return modify_frequency_device( *this, *d_mod, *d_in, device_interp.kind() );
	}

PV PV::repitch( const Function<TF, float> & factor, const Interpolator & interp ) const
	{
	if( is_null() ) return PV();
	const DeviceInterp device_interp( interp );                                      // named kind, or the callable sampled into a table
	if( !device_interp.ok() ) return PV();
	auto d_grid = function_grid_to_device( *this, factor );                        // PVModify.cpp:275
	const MF * d_pv = device_data();
	auto out = DeviceBlock::allocate( sizeof( MF ) * size_t( get_num_channels() ) * get_num_frames() * get_num_bins() );
	if( !d_grid || !d_pv || !out ) return PV();
	// :278-302 running sum over bins, bin_to_frequency, per-MF lerp, then modify_frequency_base (:196-257) -- one call, on the device
	if( !detail::report( flanhip_repitch_interp_dev( reinterpret_cast<const flanhip_MF*>( d_pv ), get_num_channels(), get_num_frames(), get_num_bins(), get_sample_rate(),
			static_cast<float*>( d_grid->ptr ), device_interp.kind(), static_cast<flanhip_MF*>( out->ptr ), nullptr ), "repitch" ) ) return PV();
	if( !detail::report( flanhip_stream_synchronize( nullptr ), "repitch" ) ) return PV();
	return PVBuffer::adopt_device( get_format(), std::move( out ) );
	}

PV PV::shape( const Function<MF, MF> & shaper, bool use_shift_alignment ) const
	{
	if( is_null() ) return PV();
	// PV.cpp:435-436: the shaper sees every MF: evaluated on the host, the placement rule runs on the device
	auto d_shaped = map_rows_to_device<MF>( *this, shaper.get_execution_policy(), [&]( int, const MF * mfs, MF * out )
		{ for( Bin bin = 0; bin < get_num_bins(); ++bin ) out[bin] = shaper( mfs[bin] ); } );
	const MF * d_pv = device_data();
	auto out = DeviceBlock::allocate( sizeof( MF ) * size_t( get_num_channels() ) * get_num_frames() * get_num_bins() );
	if( !d_shaped || !d_pv || !out ) return PV();
	// without alignment the kernel that writes the result can leave convert_to_audio's pre-pass for it in a workspace
	const size_t ws_bytes = use_shift_alignment ? 0 : flanhip_synthesize_workspace_bytes( get_num_channels(), get_num_frames(), get_num_bins(), get_sample_rate(), get_analysis_rate(), get_window_size() );
	auto ws = ws_bytes ? DeviceBlock::allocate( ws_bytes ) : nullptr;
	const int rc = ws
		? flanhip_shape_table_dev_fused( reinterpret_cast<const flanhip_MF*>( d_pv ), static_cast<const flanhip_MF*>( d_shaped->ptr ), get_num_channels(), get_num_frames(),
			get_num_bins(), get_sample_rate(), get_analysis_rate(), static_cast<flanhip_MF*>( out->ptr ), get_window_size(), ws->ptr, nullptr )
		: flanhip_shape_table_dev( reinterpret_cast<const flanhip_MF*>( d_pv ), static_cast<const flanhip_MF*>( d_shaped->ptr ), get_num_channels(),
			get_num_frames(), get_num_bins(), get_sample_rate(), use_shift_alignment ? 1 : 0, static_cast<flanhip_MF*>( out->ptr ), nullptr );
	if( !detail::report( rc, "shape" ) ) return PV();
	if( !detail::report( flanhip_stream_synchronize( nullptr ), "shape" ) ) return PV();
	PV result = PVBuffer::adopt_device( get_format(), std::move( out ) );
	if( ws ) result.attach_synthesis_workspace( std::move( ws ) );
	return result;
	}

PV PV::shape_affine( float a, float b, float c, float d, bool use_shift_alignment ) const
	{
	if( is_null() ) return PV();
	const MF * d_pv = device_data();
	auto out = DeviceBlock::allocate( sizeof( MF ) * size_t( get_num_channels() ) * get_num_frames() * get_num_bins() );
	if( !d_pv || !out ) return PV();
	const size_t ws_bytes = use_shift_alignment ? 0 : flanhip_synthesize_workspace_bytes( get_num_channels(), get_num_frames(), get_num_bins(), get_sample_rate(), get_analysis_rate(), get_window_size() );
	auto ws = ws_bytes ? DeviceBlock::allocate( ws_bytes ) : nullptr;
	const int rc = ws
		? flanhip_shape_affine_dev_fused( reinterpret_cast<const flanhip_MF*>( d_pv ), get_num_channels(), get_num_frames(), get_num_bins(), get_sample_rate(), get_analysis_rate(),
			a, b, c, d, static_cast<flanhip_MF*>( out->ptr ), get_window_size(), ws->ptr, nullptr )
		: flanhip_shape_affine_dev( reinterpret_cast<const flanhip_MF*>( d_pv ), get_num_channels(), get_num_frames(), get_num_bins(), get_sample_rate(),
			a, b, c, d, use_shift_alignment ? 1 : 0, static_cast<flanhip_MF*>( out->ptr ), nullptr );
	if( !detail::report( rc, "shape_affine" ) ) return PV();
	if( !detail::report( flanhip_stream_synchronize( nullptr ), "shape_affine" ) ) return PV();
	PV result = PVBuffer::adopt_device( get_format(), std::move( out ) );
	if( ws ) result.attach_synthesis_workspace( std::move( ws ) );
	return result;
	}

// ---------------------------------------------------------------------------------------------------------------------
// further frame processors (PV/PV.cpp:205-264, :552-641; PV/PVModify.cpp:445-511, :607-666)
// ---------------------------------------------------------------------------------------------------------------------
namespace {
// a sampled Function<TF,float> as the ( device grid or nullptr, constant ) pair the C ABI takes
struct GridArg { std::shared_ptr<DeviceBlock> block; const float * ptr = nullptr; float constant = 0.0f; bool ok = true; };
GridArg grid_arg( const FunctionSample2d<float> & s )
	{
	GridArg g;
	if( s.is_constant() ) { g.constant = s.get_constant(); return g; }
	g.block = upload( s.get_vector().data(), sizeof( float ) * s.size() );
	g.ok = bool( g.block );
	if( g.ok ) g.ptr = static_cast<const float*>( g.block->ptr );
	return g;
	}
PV finish( int rc, const char * what, const PVBuffer::Format & f, std::shared_ptr<DeviceBlock> out )
	{
	if( !detail::report( rc, what ) ) return PV();
	if( !detail::report( flanhip_stream_synchronize( nullptr ), what ) ) return PV();
	return PVBuffer::adopt_device( f, std::move( out ) );
	}
size_t mf_count( const PVBuffer::Format & f ) { return size_t( f.num_channels ) * f.num_frames * f.num_bins; }
}

static PV combine_amplitudes( const PV & me, const PV & amp_source, const Function<TF, float> & amount, bool subtract )
	{
	if( me.is_null() || amp_source.is_null() ) return PV();                        // PV.cpp:207-210
	const GridArg g = grid_arg( me.sample_function_over_domain( amount ) );        // :211
	const MF * d_pv = me.device_data();
	const MF * d_src = amp_source.device_data();
	auto out = DeviceBlock::allocate( sizeof( MF ) * mf_count( me.get_format() ) );
	if( !g.ok || !d_pv || !d_src || !out ) return PV();
	const auto fn = subtract ? flanhip_subtract_amplitudes_dev : flanhip_replace_amplitudes_dev;
	const int rc = fn( reinterpret_cast<const flanhip_MF*>( d_pv ), me.get_num_channels(), me.get_num_frames(), me.get_num_bins(),
		reinterpret_cast<const flanhip_MF*>( d_src ), amp_source.get_num_channels(), amp_source.get_num_frames(), amp_source.get_num_bins(),
		g.ptr, g.constant, static_cast<flanhip_MF*>( out->ptr ), nullptr );
	return finish( rc, subtract ? "subtract_amplitudes" : "replace_amplitudes", me.get_format(), std::move( out ) );
	}

PV PV::replace_amplitudes( const PV & amp_source, const Function<TF, float> & amount ) const { return combine_amplitudes( *this, amp_source, amount, false ); }
PV PV::subtract_amplitudes( const PV & amp_source, const Function<TF, float> & amount ) const { return combine_amplitudes( *this, amp_source, amount, true ); }

// predicateNLoudestPartials, PV.cpp:552-590
static PV n_loudest( const PV & me, const Function<Second, Bin> & num_bins, bool remove )
	{
	if( me.is_null() ) return PV();
	const auto sampled = num_bins.sample( 0, me.get_num_frames(), me.frame_to_time( 1 ) );   // :555
	std::shared_ptr<DeviceBlock> d_n;
	if( !sampled.is_constant() )
		{
		static_assert( sizeof( Bin ) == sizeof( int32_t ), "Bin is int32" );
		d_n = upload( sampled.get_vector().data(), sizeof( int32_t ) * sampled.get_vector().size() );
		if( !d_n ) return PV();
		}
	const MF * d_pv = me.device_data();
	auto out = DeviceBlock::allocate( sizeof( MF ) * mf_count( me.get_format() ) );
	if( !d_pv || !out ) return PV();
	const int rc = flanhip_n_loudest_partials_dev( reinterpret_cast<const flanhip_MF*>( d_pv ), me.get_num_channels(), me.get_num_frames(), me.get_num_bins(),
		d_n ? static_cast<const int32_t*>( d_n->ptr ) : nullptr, sampled.is_constant() ? int32_t( sampled.get_constant() ) : 0, remove ? 1 : 0,
		static_cast<flanhip_MF*>( out->ptr ), nullptr );
	return finish( rc, remove ? "remove_n_loudest_partials" : "retain_n_loudest_partials", me.get_format(), std::move( out ) );
	}

PV PV::retain_n_loudest_partials( const Function<Second, Bin> & num_bins ) const { return n_loudest( *this, num_bins, false ); }
PV PV::remove_n_loudest_partials( const Function<Second, Bin> & num_bins ) const { return n_loudest( *this, num_bins, true ); }

PV PV::resonate( Second length, const Function<TF, float> & decay ) const
	{
	if( is_null() ) return PV();
	const int64_t Fo = flanhip_resonate_out_frames( get_num_frames(), length, get_sample_rate(), get_hop_size() );   // PV.cpp:609-613
	if( Fo < get_num_frames() ) return PV();
	PVBuffer::Format f = get_format();
	f.num_frames = Frame( Fo );
	// :616: decay is sampled over the OUTPUT's domain
	const GridArg g = grid_arg( decay.sample( 0, float( f.num_frames ), 1.0f / get_analysis_rate(), 0, float( f.num_bins ), bin_to_frequency( 1 ) ) );
	const MF * d_pv = device_data();
	auto out = DeviceBlock::allocate( sizeof( MF ) * mf_count( f ) );
	if( !g.ok || !d_pv || !out ) return PV();
	const int rc = flanhip_resonate_dev( reinterpret_cast<const flanhip_MF*>( d_pv ), get_num_channels(), get_num_frames(), get_num_bins(), get_sample_rate(),
		get_hop_size(), Fo, g.ptr, g.constant, static_cast<flanhip_MF*>( out->ptr ), nullptr );
	return finish( rc, "resonate", f, std::move( out ) );
	}

PV PV::desample( const Function<TF, float> & decimation_ratio, const Interpolator & interp ) const
	{
	if( is_null() ) return PV();
	const DeviceInterp device_interp( interp );                                      // named kind, or the callable sampled into a table
	if( !device_interp.ok() ) return PV();
	const GridArg g = grid_arg( sample_function_over_domain( decimation_ratio ) ); // PVModify.cpp:450
	const MF * d_pv = device_data();
	auto out = DeviceBlock::allocate( sizeof( MF ) * mf_count( get_format() ) );
	if( !g.ok || !d_pv || !out ) return PV();
	const int rc = flanhip_desample_dev( reinterpret_cast<const flanhip_MF*>( d_pv ), get_num_channels(), get_num_frames(), get_num_bins(), g.ptr, g.constant,
		device_interp.kind(), static_cast<flanhip_MF*>( out->ptr ), nullptr );
	return finish( rc, "desample", get_format(), std::move( out ) );
	}

PV PV::time_extrapolate( Second start_time, Second end_time, Second extrapolation_time, const Interpolator & interpolator ) const
	{
	if( is_null() ) return PV();
	// Input validation, PVModify.cpp:612-617
	start_time = std::clamp( start_time, 0.0f, get_length() );
	if( end_time == -1 ) end_time = get_length();
	end_time = std::clamp( end_time, 0.0f, get_length() );
	if( start_time >= end_time ) return PV();
	if( extrapolation_time <= 0 ) return PV();
	const Frame start_frame = Frame( time_to_frame( start_time ) );                // :619-621
	Frame end_frame = Frame( time_to_frame( end_time ) );
	const Frame ext_frames = Frame( time_to_frame( extrapolation_time ) );
	end_frame = std::min( end_frame, get_num_frames() - 1 );                       // the reference reads this frame unchecked (:649)
	if( start_frame >= end_frame || ext_frames <= 0 ) return PV();
	PVBuffer::Format f = get_format();
	f.num_frames = end_frame + ext_frames;                                         // :623-624
	std::vector<float> interp_samples( size_t( f.num_frames - start_frame ) );     // :631-633, literally
	for( Frame frame = 0; frame < Frame( interp_samples.size() ); ++frame )
		interp_samples[frame] = interpolator( float( frame - start_frame ) / float( end_frame - start_frame ) );
	auto d_samples = upload( interp_samples.data(), sizeof( float ) * interp_samples.size() );
	const MF * d_pv = device_data();
	auto out = DeviceBlock::allocate( sizeof( MF ) * mf_count( f ) );
	if( !d_samples || !d_pv || !out ) return PV();
	const int rc = flanhip_time_extrapolate_dev( reinterpret_cast<const flanhip_MF*>( d_pv ), get_num_channels(), get_num_frames(), get_num_bins(),
		get_sample_rate(), start_frame, end_frame, f.num_frames, static_cast<const float*>( d_samples->ptr ), static_cast<flanhip_MF*>( out->ptr ), nullptr );
	return finish( rc, "time_extrapolate", f, std::move( out ) );
	}

// ---------------------------------------------------------------------------------------------------------------------
// selecting, rearranging and re-placing frames and bins (PV/PV.cpp:24-39, :92-198, :362-419, :643-727)
// ---------------------------------------------------------------------------------------------------------------------
PV PV::get_frame( Second time ) const
	{
	if( is_null() ) return PV();
	const float selected = std::clamp( time_to_frame( time ), 0.0f, float( get_num_frames() - 1 ) );   // PV.cpp:28
	PVBuffer::Format f = get_format();
	f.num_frames = 1;                                                               // :30-31
	const MF * d_pv = device_data();
	auto out = DeviceBlock::allocate( sizeof( MF ) * mf_count( f ) );
	if( !d_pv || !out ) return PV();
	const int rc = flanhip_get_frame_dev( reinterpret_cast<const flanhip_MF*>( d_pv ), get_num_channels(), get_num_frames(), get_num_bins(), selected,
		FLANHIP_INTERP_LINEAR, static_cast<flanhip_MF*>( out->ptr ), nullptr );
	return finish( rc, "get_frame", f, std::move( out ) );
	}

PV PV::select( Second length, const Function<TF, TF> & selector ) const
	{
	if( is_null() ) return PV();
	if( length <= 0.0f ) return PV();                                               // PV.cpp:98
	PVBuffer::Format f = get_format();
	f.num_frames = Frame( time_to_frame( length ) );                                // :100-101
	if( f.num_frames <= 0 ) return PV();
	auto d_sel = function_grid_to_device( *this, selector, static_cast<FunctionSample2d<TF>*>( nullptr ), f.num_frames );   // :103, over the OUTPUT's domain
	const MF * d_pv = device_data();
	auto out = DeviceBlock::allocate( sizeof( MF ) * mf_count( f ) );
	if( !d_sel || !d_pv || !out ) return PV();
	const int rc = flanhip_select_dev( reinterpret_cast<const flanhip_MF*>( d_pv ), get_num_channels(), get_num_frames(), get_num_bins(), get_sample_rate(),
		get_hop_size(), static_cast<const float*>( d_sel->ptr ), f.num_frames, static_cast<flanhip_MF*>( out->ptr ), nullptr );
	return finish( rc, "select", f, std::move( out ) );
	}

PV PV::freeze( const std::vector<Second> & pause_times, const std::vector<Second> & pause_lengths ) const
	{
	if( is_null() ) return PV();
	if( pause_lengths.size() != pause_times.size() )                                // PV.cpp:135-139
		{
		std::cerr << "Error in flan::PV::freeze: pause_times and pause_lengths were not the same size.";
		return PV();
		}
	const int n = int( pause_times.size() );
	const int64_t Fo = flanhip_freeze_plan( get_num_frames(), get_sample_rate(), get_hop_size(), pause_times.data(), pause_lengths.data(), n, nullptr );
	if( Fo <= 0 ) return PV();
	detail::StagingVector<int32_t> src( static_cast<size_t>( Fo ) );
	flanhip_freeze_plan( get_num_frames(), get_sample_rate(), get_hop_size(), pause_times.data(), pause_lengths.data(), n, src.data() );
	PVBuffer::Format f = get_format();
	f.num_frames = Frame( Fo );                                                     // :170-171
	auto d_src = upload( src.data(), sizeof( int32_t ) * src.size() );
	const MF * d_pv = device_data();
	auto out = DeviceBlock::allocate( sizeof( MF ) * mf_count( f ) );
	if( !d_src || !d_pv || !out ) return PV();
	const int rc = flanhip_select_frames_dev( reinterpret_cast<const flanhip_MF*>( d_pv ), get_num_channels(), get_num_frames(), get_num_bins(),
		static_cast<const int32_t*>( d_src->ptr ), Fo, static_cast<flanhip_MF*>( out->ptr ), nullptr );
	return finish( rc, "freeze", f, std::move( out ) );
	}

PV PV::modify( const Function<TF, TF> & mod, const Interpolator & interp ) const
	{
	if( is_null() ) return PV();
	const DeviceInterp device_interp( interp );                                      // named kind, or the callable sampled into a table
	if( !device_interp.ok() ) return PV();
	// PVModify.cpp:22: mod over this PV's grid -- on the host for the output's length (:28-38), on the device for the kernels
	FunctionSample2d<TF> sampled{ TF{ 0.0f, 0.0f }, 0, 0 };
	auto d_mod = function_grid_to_device( *this, mod, &sampled );
	if( !d_mod ) return PV();
	detail::StagingVector<TF> filled;
	if( sampled.is_constant() ) filled.assign( sampled.size(), sampled.get_constant() );
	const TF * host_grid = sampled.is_constant() ? filled.data() : sampled.get_vector().data();
	const int64_t Fo = flanhip_modify_out_frames( reinterpret_cast<const float*>( host_grid ), get_num_frames(), get_num_bins(), get_sample_rate(), get_hop_size() );
	if( Fo == -2 )                                                                  // :30-34
		{
		std::cout << "PV::modify tried to make a file longer than 10 minutes, which is currently disabled";
		return PV();
		}
	if( Fo <= 0 ) return PV();
	// :62-66: mod again, at every MF's own frequency -- data dependent, so on the host
	auto d_in_f = map_rows_to_device<float>( *this, mod.get_execution_policy(), [&]( int row, const MF * mfs, float * out )
		{
		const Second t = frame_to_time( fFrame( Frame( row % get_num_frames() ) ) );
		for( Bin bin = 0; bin < get_num_bins(); ++bin ) out[bin] = mod( TF{ t, mfs[bin].f } ).f;
		} );
	PVBuffer::Format f = get_format();
	f.num_frames = Frame( Fo );
	const MF * d_pv = device_data();
	auto out = DeviceBlock::allocate( sizeof( MF ) * mf_count( f ) );
	if( !d_in_f || !d_pv || !out ) return PV();
	const int rc = flanhip_modify_dev( reinterpret_cast<const flanhip_MF*>( d_pv ), get_num_channels(), get_num_frames(), get_num_bins(), get_sample_rate(), get_hop_size(),
		static_cast<const float*>( d_mod->ptr ), static_cast<const float*>( d_in_f->ptr ), device_interp.kind(), Fo, static_cast<flanhip_MF*>( out->ptr ), nullptr );
	return finish( rc, "modify", f, std::move( out ) );
	}

PV PV::stretch_spline( const Function<Second, float> & interpolation ) const
	{
	if( is_null() ) return PV();
	if( get_num_frames() < 3 ) { std::cerr << "flan: stretch_spline needs at least three frames" << std::endl; return PV(); }   // spline.h:288 asserts it
	// PVModify.cpp:391-394 safeInterpolation; the float -> uint32 conversion (undefined for negatives and NaN there) saturates here
	std::vector<uint32_t> steps( size_t( get_num_frames() - 1 ) );
	const float seconds_per_frame = frame_to_time( 1 );
	for( Frame frame = 0; frame + 1 < get_num_frames(); ++frame )
		{
		const float v = interpolation( frame * seconds_per_frame );
		const uint32_t u = !( v >= 1.0f ) ? 0u : ( v >= 4294967296.0f ? 0xFFFFFFFFu : uint32_t( v ) );
		steps[size_t( frame )] = std::max( u, 1u );
		}
	const int64_t Fo = flanhip_stretch_spline_out_frames( steps.data(), get_num_frames() );   // :399-405
	if( Fo <= 0 ) return PV();
	PVBuffer::Format f = get_format();
	f.num_frames = Frame( Fo );
	const MF * d_pv = device_data();
	auto out = DeviceBlock::allocate( sizeof( MF ) * mf_count( f ) );
	if( !d_pv || !out ) return PV();
	const int rc = flanhip_stretch_spline_dev( reinterpret_cast<const flanhip_MF*>( d_pv ), get_num_channels(), get_num_frames(), get_num_bins(), steps.data(), Fo,
		static_cast<flanhip_MF*>( out->ptr ), nullptr );
	return finish( rc, "stretch_spline", f, std::move( out ) );
	}

PV PV::smear_time( const Function<TF, Second> & smear_size, const Function<TF, int> & granularity, const Function<Second, float> & distribution ) const
	{
	if( is_null() ) return PV();
	// PVModify.cpp:520-524: both grids over this PV's domain (the clamps to >= 1 and >= 0 happen where the values are used)
	FunctionSample2d<Second> smear_host{ 0.0f, 0, 0 };
	std::shared_ptr<DeviceBlock> d_smear, d_gran;
	if( !smear_size.is_constant() ) { d_smear = function_grid_to_device( *this, smear_size, &smear_host ); if( !d_smear ) return PV(); }
	if( !granularity.is_constant() ) { d_gran = function_grid_to_device( *this, granularity ); if( !d_gran ) return PV(); }
	int32_t true_left = 0, dist_samples_2 = 0;
	int64_t Fo = 0;
	if( !detail::report( flanhip_smear_time_plan( get_num_frames(), get_num_bins(), get_sample_rate(), get_hop_size(),
			smear_size.is_constant() ? nullptr : smear_host.get_vector().data(), smear_size.is_constant() ? smear_size.get_constant() : 0.0f,
			&true_left, &Fo, &dist_samples_2 ), "smear_time" ) ) return PV();       // :536-564
	if( Fo <= 0 ) return PV();
	// :558-560: the distribution sampled on [-1, 1)
	std::vector<float> dist;
	if( dist_samples_2 > 0 )
		{
		const auto sampled = distribution.sample( -dist_samples_2, dist_samples_2, 1.0f / dist_samples_2 );
		if( sampled.is_constant() ) dist.assign( size_t( 2 ) * dist_samples_2, sampled.get_constant() );
		else dist = sampled.get_vector();
		}
	auto d_dist = dist.empty() ? nullptr : upload( dist.data(), sizeof( float ) * dist.size() );
	PVBuffer::Format f = get_format();
	f.num_frames = Frame( Fo );                                                     // :562-563
	const MF * d_pv = device_data();
	auto out = DeviceBlock::allocate( sizeof( MF ) * mf_count( f ) );
	if( ( !dist.empty() && !d_dist ) || !d_pv || !out ) return PV();
	const int rc = flanhip_smear_time_dev( reinterpret_cast<const flanhip_MF*>( d_pv ), get_num_channels(), get_num_frames(), get_num_bins(), get_sample_rate(), get_hop_size(),
		d_smear ? static_cast<const float*>( d_smear->ptr ) : nullptr, smear_size.is_constant() ? smear_size.get_constant() : 0.0f,
		d_gran ? static_cast<const int32_t*>( d_gran->ptr ) : nullptr, granularity.is_constant() ? granularity.get_constant() : 0,
		d_dist ? static_cast<const float*>( d_dist->ptr ) : nullptr, int64_t( dist.size() ), true_left, Fo, static_cast<flanhip_MF*>( out->ptr ), nullptr );
	return finish( rc, "smear_time", f, std::move( out ) );
	}

// harmonic_scaler, PV.cpp:362-407
static PV harmonic_scale( const PV & me, const Function<std::pair<Second, Harmonic>, float> & series, int mode, Harmonic num_harmonics )
	{
	const Frame frames = me.get_num_frames();
	const size_t H = size_t( std::max( num_harmonics, 0 ) );
	detail::StagingVector<float> sampled( H * frames );
	detail::for_each_index( 0, frames, series.get_execution_policy(), [&]( int frame )   // :371-379: the callable sees the 0-based harmonic index
		{
		const Second t = me.frame_to_time( fFrame( frame ) );
		for( size_t h = 0; h < H; ++h ) sampled[h + size_t( frame ) * H] = series( std::pair<Second, Harmonic>( t, Harmonic( h ) ) );
		}, 16 );
	auto d_series = H ? upload( sampled.data(), sizeof( float ) * sampled.size() ) : nullptr;
	const MF * d_pv = me.device_data();
	auto out = DeviceBlock::allocate( sizeof( MF ) * mf_count( me.get_format() ) );
	if( ( H && !d_series ) || !d_pv || !out ) return PV();
	const int rc = flanhip_harmonic_scale_dev( reinterpret_cast<const flanhip_MF*>( d_pv ), me.get_num_channels(), frames, me.get_num_bins(), me.get_sample_rate(),
		d_series ? static_cast<const float*>( d_series->ptr ) : nullptr, int( H ), mode, static_cast<flanhip_MF*>( out->ptr ), nullptr );
	return finish( rc, mode == 0 ? "add_octaves" : "add_harmonics", me.get_format(), std::move( out ) );
	}

PV PV::add_octaves( const Function<std::pair<Second, Harmonic>, float> & series ) const
	{
	if( is_null() ) return PV();
	return harmonic_scale( *this, series, 0, Harmonic( std::ceil( std::log2( get_height() ) ) ) );   // PV.cpp:412
	}

PV PV::add_harmonics( const Function<std::pair<Second, Harmonic>, float> & series ) const
	{
	if( is_null() ) return PV();
	return harmonic_scale( *this, series, 1, get_num_bins() );                      // PV.cpp:418
	}

PV PV::cut_frames( Frame start, Frame end ) const
	{
	if( is_null() ) return PV::create_null();
	int32_t first = 0, count = 0;
	flanhip_cut_frames_range( get_num_frames(), start, end, &first, &count );       // PV.cpp:651-653
	if( count <= 0 ) return PV::create_null();                                      // (an empty range gives an empty = null PV there too)
	PVBuffer::Format f = get_format();
	f.num_frames = count;
	const MF * d_pv = device_data();
	auto out = DeviceBlock::allocate( sizeof( MF ) * mf_count( f ) );
	if( !d_pv || !out ) return PV();
	const int rc = flanhip_cut_frames_dev( reinterpret_cast<const flanhip_MF*>( d_pv ), get_num_channels(), get_num_frames(), get_num_bins(), first, count,
		static_cast<flanhip_MF*>( out->ptr ), nullptr );
	return finish( rc, "cut_frames", f, std::move( out ) );
	}

std::vector<PV> PV::split_at_times( std::vector<Second> split_times ) const
	{
	if( is_null() ) return std::vector<PV>();
	std::sort( split_times.begin(), split_times.end() );                            // PV.cpp:676
	std::vector<Frame> split_frames;
	split_frames.push_back( 0 );
	for( Second t : split_times )                                                   // :680-686
		{
		const Frame f = Frame( time_to_frame( t ) );
		if( f <= 0 ) continue;
		if( get_num_frames() <= f ) break;
		split_frames.push_back( f );
		}
	split_frames.push_back( get_num_frames() );
	std::vector<PV> outs;
	outs.reserve( split_frames.size() - 1 );
	for( size_t i = 0; i + 1 < split_frames.size(); ++i ) outs.push_back( cut_frames( split_frames[i], split_frames[i + 1] ) );   // :690-693
	return outs;
	}

PV PV::join( const std::vector<const PV *> & ins )
	{
	if( ins.size() == 0 ) return PV::create_null();
	PVBuffer::Format f = ins[0]->get_format();                                      // PV.cpp:704
	int64_t total = 0;
	for( const PV * x : ins ) total += x->get_num_frames();                         // :705
	if( total <= 0 || total > INT32_MAX ) return PV::create_null();
	f.num_frames = Frame( total );
	auto out = DeviceBlock::allocate( sizeof( MF ) * mf_count( f ) );
	if( !out ) return PV();
	if( !detail::report( flanhip_memset( out->ptr, 0, sizeof( MF ) * mf_count( f ), nullptr ), "join" ) ) return PV();   // :706-707
	int64_t at = 0;
	for( const PV * x : ins )                                                       // :709-717
		{
		if( x->get_num_frames() <= 0 || x->get_num_channels() <= 0 || x->get_num_bins() <= 0 ) continue;
		const MF * d_in = x->device_data();
		if( !d_in ) return PV();
		if( !detail::report( flanhip_place_frames_dev( reinterpret_cast<const flanhip_MF*>( d_in ), x->get_num_channels(), x->get_num_frames(), x->get_num_bins(),
				static_cast<flanhip_MF*>( out->ptr ), f.num_channels, f.num_frames, f.num_bins, at, nullptr ), "join" ) ) return PV();
		at += x->get_num_frames();
		}
	return finish( FLANHIP_OK, "join", f, std::move( out ) );
	}

PV PV::join( const std::vector<PV> & ins )
	{
	std::vector<const PV *> ptrs( ins.size() );                                     // PV.cpp:16-22
	std::transform( ins.begin(), ins.end(), ptrs.begin(), []( const PV & p ){ return &p; } );
	return join( ptrs );
	}

} // namespace flan
