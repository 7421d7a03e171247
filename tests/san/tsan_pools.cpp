// tsan_pools.cpp -- the library's host thread pools (libdwt_amd/csrc/dwt_host_pools.h) under ThreadSanitizer with
// HOST-ONLY jobs: the RowPool as the host-pointer calls use it (several caller threads taking turns, jobs of many
// row chunks writing disjoint rows) and the SlotThread workers as the multi-GPU entries use them (submit / wait
// rounds, a failing job's message read on the worker).  Any report makes the run fail (halt_on_error).
#include "dwt_host_pools.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>

using namespace dwtb;

static thread_local char t_err[64] = "";
static const char *last_error() { return t_err; }

int main()
{
	// ---- RowPool: three caller threads, each repacking "images" of different sizes a few dozen times ----
	std::atomic<long> checksum{0};
	auto caller = [&](int id) {
		for (int rep = 0; rep < 40; rep++) {
			const int rows = 64 + 37 * id + rep, w = 257 + id;
			std::vector<int> src((size_t)rows * w), dst((size_t)rows * w, -1);
			for (size_t i = 0; i < src.size(); i++)
				src[i] = (int)(i * 2654435761u) ^ id;
			RowPool::get().run(rows, 8, [&](int r0, int r1) {
				for (int y = r0; y < r1; y++)
					memcpy(&dst[(size_t)y * w], &src[(size_t)y * w], (size_t)w * sizeof(int));
			});
			if (memcmp(src.data(), dst.data(), src.size() * sizeof(int))) {
				fprintf(stderr, "RowPool: rows missing (caller %d rep %d)\n", id, rep);
				exit(3);
			}
			checksum += dst[dst.size() / 2];
		}
	};
	std::vector<std::thread> callers;
	for (int id = 0; id < 3; id++)
		callers.emplace_back(caller, id);
	for (auto &t : callers)
		t.join();

	// ---- SlotThread: four workers, rounds of submit / wait, every third job failing with a message ----
	std::vector<SlotThread *> slots;
	for (int k = 0; k < 4; k++)
		slots.push_back(new SlotThread(last_error));
	int results[4] = {0, 0, 0, 0};
	for (int round = 0; round < 200; round++) {
		for (int k = 0; k < 4; k++) {
			int *out = &results[k];
			slots[k]->submit([=] {
				*out = round * 4 + k;
				if ((round + k) % 3 == 0) {
					snprintf(t_err, sizeof t_err, "job %d of slot %d failed", round, k);
					return 1;
				}
				return 0;
			});
		}
		for (int k = 0; k < 4; k++) {
			std::string err;
			const int rc = slots[k]->wait(err);
			const bool should_fail = (round + k) % 3 == 0;
			char want[64];
			snprintf(want, sizeof want, "job %d of slot %d failed", round, k);
			if (rc != (should_fail ? 1 : 0) || results[k] != round * 4 + k || (should_fail && err != want) || (!should_fail && !err.empty())) {
				fprintf(stderr, "SlotThread: round %d slot %d: rc %d result %d err '%s'\n", round, k, rc, results[k], err.c_str());
				exit(4);
			}
		}
	}
	printf("tsan_pools OK (%ld)\n", checksum.load());
	return 0;
}
