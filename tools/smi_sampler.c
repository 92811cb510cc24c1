/* Clock / power sampler for tools/power_probe.py (test tooling, not product code).
 *   smi_sampler <out.csv> <period_ms> <max_seconds>
 * Every period: the SMU's gpu_metrics table through librocm_smi64 (current_gfxclk, the per-XCD current_gfxclks[8],
 * current_uclk, current / average socket power, hotspot temperature, throttle status), one CSV line each, until SIGTERM or
 * max_seconds.  Build: gcc -O2 tools/smi_sampler.c -I/opt/rocm/include -L/opt/rocm/lib -lrocm_smi64 -Wl,-rpath,/opt/rocm/lib */
#include <rocm_smi/rocm_smi.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static volatile int g_stop = 0;
static void on_term(int s) { (void)s; g_stop = 1; }
static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}
int main(int argc, char** argv) {
    if (argc < 4) return 2;
    FILE* f = fopen(argv[1], "w");
    if (!f) return 3;
    const double period = atof(argv[2]) * 1e-3, max_s = atof(argv[3]);
    signal(SIGTERM, on_term);
    signal(SIGINT, on_term);
    rsmi_status_t st = rsmi_init(0);
    if (st != RSMI_STATUS_SUCCESS) { fprintf(f, "# rsmi_init failed: %d\n", (int)st); fclose(f); return 4; }
    fprintf(f, "t_s,gfxclk_mhz,uclk_mhz,cur_socket_w,avg_socket_w,hotspot_c,throttle,gfx_activity,xcd0,xcd1,xcd2,xcd3,xcd4,xcd5,xcd6,xcd7,rsmi_sclk_mhz,rsmi_power_w\n");
    const double t0 = now_s();
    while (!g_stop && now_s() - t0 < max_s) {
        const double t = now_s();
        rsmi_gpu_metrics_t m;
        memset(&m, 0, sizeof(m));
        st = rsmi_dev_gpu_metrics_info_get(0, &m);
        rsmi_frequencies_t fr;
        memset(&fr, 0, sizeof(fr));
        double sclk = -1, pw = -1;
        if (rsmi_dev_gpu_clk_freq_get(0, RSMI_CLK_TYPE_SYS, &fr) == RSMI_STATUS_SUCCESS && fr.current < RSMI_MAX_NUM_FREQUENCIES)
            sclk = fr.frequency[fr.current] * 1e-6;
        uint64_t p = 0;
        RSMI_POWER_TYPE pt;
        if (rsmi_dev_power_get(0, &p, &pt) == RSMI_STATUS_SUCCESS) pw = p * 1e-6;
        if (st == RSMI_STATUS_SUCCESS) {
            fprintf(f, "%.4f,%u,%u,%u,%u,%u,%u,%u", t - t0, m.current_gfxclk, m.current_uclk, m.current_socket_power, m.average_socket_power,
                    m.temperature_hotspot, m.throttle_status, m.average_gfx_activity);
            for (int i = 0; i < 8; i++) fprintf(f, ",%u", m.current_gfxclks[i]);
        } else {
            fprintf(f, "%.4f,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1", t - t0);
        }
        fprintf(f, ",%.1f,%.1f\n", sclk, pw);
        fflush(f);
        const double left = period - (now_s() - t);
        if (left > 0) {
            struct timespec sl = {(time_t)left, (long)((left - (time_t)left) * 1e9)};
            nanosleep(&sl, NULL);
        }
    }
    rsmi_shut_down();
    fclose(f);
    return 0;
}
