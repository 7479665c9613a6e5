// vs_train_device_sites.h — numbering of the dropout modules, shared by host launch code and device kernels.
#pragma once

// dropout sites (the `site` of drop_site): 0 = embedding (PositionalEncoding.dropout, p = sparsity);
// layer l: 1 + 4l + {0: attention weights, 1: dropout1, 2: mlp.dropout, 3: dropout2}
#define VS_SITE_EMBED 0u
#define VS_SITE_LAYER(l, which) (1u + 4u * (unsigned)(l) + (unsigned)(which))
#define VS_SITE_ATTN 0
#define VS_SITE_DROP1 1
#define VS_SITE_MLP 2
#define VS_SITE_DROP2 3
