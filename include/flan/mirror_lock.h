// flan/mirror_lock.h -- the lock that guards an object's lazily mirrored state (host copy <-> HBM copy, attached workspaces).
//
// The reference's const methods are pure reads and may run concurrently on one object; here a const method may bring the host copy
// over from the device, upload it, or hand a workspace on, so those steps take this per-object mutex.  The buffers are move-only
// with defaulted moves: a moved-to object simply gets a fresh, unlocked mutex (nobody may be using either object during a move).
#pragma once
#include <mutex>

namespace flan { namespace detail {

struct MirrorLock
	{
	mutable std::mutex m;
	MirrorLock() = default;
	MirrorLock( const MirrorLock & ) {}
	MirrorLock( MirrorLock && ) noexcept {}
	MirrorLock & operator=( const MirrorLock & ) { return *this; }
	MirrorLock & operator=( MirrorLock && ) noexcept { return *this; }
	std::unique_lock<std::mutex> hold() const { return std::unique_lock<std::mutex>( m ); }
	};

} }
