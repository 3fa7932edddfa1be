// integration/bb_de_header_gpu.cpp -- the body of
//     void bb_de_header::execute(int _plp_id, l1_postsignalling _l1_post, int _len_in, uint8_t* _in)
//     (/root/reference/src/DVB_T2/bb_de_header.h:59; the reference's body: bb_de_header.cpp:84-448)
// Sequential host code in the library as in the reference (BBHEADER, CRC-8 / mode, SYNCD, HEM and NM packet assembly across BBFRAMEs,
// need_plp); what comes back is the run of transport-stream bytes the reference would have written for this BBFRAME, and it goes out the
// way the reference sends it: one UDP datagram or one write to the file (:433-443). Messages as the reference words them.
// (The text block set_info builds for the GUI once per PLP, :452-498, is not produced by this body.)
#include <QDataStream>
#include <QFile>
#include <QHostAddress>
#include <QUdpSocket>

#include "bb_de_header.h"          // the reference's
#include "t2gpu_ref_glue.h"

namespace {
t2gpu_bbdh *dh = nullptr;
int dh_need_plp = -1;
}

void bb_de_header::execute(int _plp_id, l1_postsignalling _l1_post, int _len_in, uint8_t* _in)
{
    mutex_in->lock();
    signal_in->wakeOne();
    (void)_l1_post;
    if (!dh || dh_need_plp != need_plp) {                 // set_out (:500-525) may have changed the PLP that is wanted
        if (dh) t2gpu_bbdh_destroy(dh);
        dh = t2gpu_bbdh_create(need_plp);
        dh_need_plp = need_plp;
        if (!dh) { t2glue::complain("t2gpu_bbdh_create"); mutex_in->unlock(); return; }
    }
    int errors = 0;
    const int n = t2gpu_bbdh_execute(dh, _plp_id, _len_in, _in, reinterpret_cast<uint8_t *>(buffer_out), len, &errors);
    if (n == -1) emit ts_stage("Baseband header CRC8 error.");
    if (n > 0) {
        switch (id_current_out) {
        case out_network:
            socket->writeDatagram(buffer_out, n, QHostAddress::LocalHost, num_port_udp);
            break;
        case out_file:
            stream->writeRawData(buffer_out, n);
            break;
        }
    }
    if (errors != 0) emit ts_stage("TS error.");
    mutex_in->unlock();
}
